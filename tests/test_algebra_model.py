"""The kernel algebra (tests/algebra_model.py) must equal the layer-by-layer oracle in fp64."""
import numpy as np
import pytest

from oracle import explainn_oracle as orc
import algebra_model as am


@pytest.mark.parametrize("U,k,L,T,B,nfrac,drop", [
    (3, 5, 26, 2, 8, 0.0, False),
    (5, 19, 61, 3, 24, 0.05, False),
    (4, 19, 75, 1, 16, 0.0, True),
    (2, 7, 40, 2, 12, 0.1, True),
])
def test_algebra_equals_oracle_fp64(U, k, L, T, B, nfrac, drop):
    sd = orc.random_state_dict(U, k, L, T, seed=3, dtype=np.float64)
    sd["linears.1.weight"][::2] *= -1            # negative gamma1 -> min-pooling branch
    x = orc.random_onehot(B, L, seed=4, n_frac=nfrac, dtype=np.float64)
    rng = np.random.default_rng(5)
    y = (rng.random((B, T)) > 0.5).astype(np.float64)
    keep = (rng.random((B, 100 * U)) > 0.3).astype(np.float64) if drop else None
    logits, cache, nb = orc.forward(sd, x, training=True, dropout_mask=keep, dtype=np.float64,
                                    return_cache=True)
    _, dl = orc.bce_with_logits(logits, y)
    grads = orc.backward(cache, dl)
    lg2, gr2, new2 = am.train_forward_backward(sd, x, lambda lg: orc.bce_with_logits(lg, y)[1],
                                               keep=keep)
    assert np.abs(logits - lg2).max() < 1e-10
    for key, v in grads.items():
        err = np.abs(v.reshape(-1) - gr2[key].reshape(-1)).max()
        assert err < 1e-9 * max(1.0, np.abs(v).max()), (key, err)
    for key, v in new2.items():
        assert np.abs(nb[key] - v).max() < 1e-10, key
