"""CPU: the filter -> PWM restatement (oracle/interpret_oracle.py) and the product's host-side
selection logic against the fixtures whose float16 activations / outputs / predictions came from
the imported reference model (tools/make_golden.py pfm)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import interpret_oracle as io

PFM_CASES = ["pfm_u8_k9", "pfm_u8_k9_rc", "pfm_u6_k19_cap", "pfm_u5_k7_lin"]


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    U, k, L, T, N, rc, lin, cap = [int(v) for v in z["meta"]]
    return z, dict(U=U, k=k, L=L, T=T, N=N, rc=bool(rc), kind="linear" if lin else "binary", cap=cap)


def onehot(codes):
    x = np.zeros((codes.shape[0], 4, codes.shape[1]), dtype=np.float32)
    for a in range(4):
        x[:, a, :] = (codes == a)
    return x


@pytest.mark.parametrize("name", PFM_CASES)
def test_oracle_activations_match_reference_float16(name):
    """The oracle's own eval forward, stored as float16 like test.py:137, against the reference's:
    equal up to one float16 ulp (a 1e-7 relative difference can straddle a rounding boundary)."""
    z, m = load(name)
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd/")}
    acts, outs, preds = io.acts_outs_preds(sd, onehot(z["codes"]))
    for mine, ref, what in ((acts, z["acts"], "acts"), (outs, z["outs"], "outs"), (preds, z["preds"], "preds")):
        a, b = mine.astype(np.float64), ref.astype(np.float64)
        ulp = np.maximum(np.abs(b), 2.0 ** -14) * 2.0 ** -10
        assert (np.abs(a - b) <= ulp).all(), what
        assert (mine == ref).mean() > 0.995, what


@pytest.mark.parametrize("name", PFM_CASES)
def test_oracle_bookkeeping_reproduces_fixture(name):
    z, m = load(name)
    idxs = io.well_predicted_sequences(z["preds"], z["labels"], m["kind"], m["rc"])
    assert np.array_equal(idxs, z["idxs"])
    thr = io.act_thresholds(z["acts"], idxs, m["rc"])
    assert thr.dtype == np.float16 and np.array_equal(thr, z["thresholds"])
    pfm, nsites = io.site_pfms(z["codes"], z["acts"], idxs, thr, m["k"], m["rc"], cap=m["cap"])
    assert np.array_equal(pfm, z["pfm"]) and np.array_equal(nsites, z["nsites"])
    # every uncapped site contributes one letter per column unless it is an N
    assert (pfm.sum(axis=2) <= nsites[:, None]).all()
    if name == "pfm_u6_k19_cap":
        assert (nsites == m["cap"]).all()


@pytest.mark.parametrize("name", PFM_CASES)
def test_product_selection_logic_matches(name):
    """explainn_amd.interpret._get_well_predicted_sequences is host logic (no device needed)."""
    from explainn_amd import interpret as it
    z, m = load(name)
    idxs = it._get_well_predicted_sequences(z["preds"], z["labels"], m["kind"], m["rc"])
    assert np.array_equal(np.asarray(idxs), z["idxs"])


def test_importances_follow_hit_matrix():
    from explainn_amd import interpret as it
    z, m = load("pfm_u8_k9")
    hit = (z["acts"] > z["thresholds"][None, :, None]).any(axis=2)
    sd_w = z["sd/final.weight"]
    res = it.filter_importances(z["outs"], sd_w, z["idxs"], hit)
    for u, (sel, imps) in enumerate(res):
        assert np.array_equal(sel, z["imp_sel/%d" % u])
        assert np.array_equal(imps, z["imp/%d" % u])


def test_jaspar_layout():
    from explainn_amd import interpret as it
    txt = it.format_jaspar(np.array([[1, 2, 3, 4], [10, 0, 0, 0]]), "filter0", "demo")
    assert txt.splitlines() == [">filter0 demo", "A [  1.00  10.00]", "C [  2.00   0.00]",
                                "G [  3.00   0.00]", "T [  4.00   0.00]"]


@pytest.mark.parametrize("scoring", ["max", "sum"])
def test_pwm_scan_oracle_matches_reference(scoring):
    from oracle import explainn_oracle as eo
    z = np.load(os.path.join(GOLDEN, "pwm_scan.npz"), allow_pickle=False)
    got = eo.pwm_scan(z["pwms"], z["x"], scoring)
    assert np.abs(got - z[scoring]).max() <= 1e-4 * max(1.0, np.abs(z[scoring]).max())
