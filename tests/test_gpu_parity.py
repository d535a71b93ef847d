"""Parity of the HIP path (through the C ABI, via the drop-in module) against the golden vectors
the imported reference produced and against the numpy oracle.  Needs an MI355X: -m gpu.

Tolerance: 1e-4 absolute on logits (BASELINE.json north_star: "within 1e-4 fp32"); gradients and
BatchNorm buffers RELATIVE to the largest entry of the reference tensor: 2e-5 against the
reference's own numbers (golden fixtures), 5e-5 against the fp64 oracle on random cases.  Tensors
with identically-zero true gradient (pre-BatchNorm biases) are checked with an absolute bound only
(SURVEY.md 7.2).  Every comparison's error/bound ratio is appended to gpurun_out/parity_margins.txt
(conftest.record_margin) so the headroom of each bound is on record."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from conftest import Golden, record_margin  # noqa: E402
from oracle import explainn_oracle as orc  # noqa: E402
from parity_util import compare_grads, reference_fp32_error  # noqa: E402

pytestmark = pytest.mark.gpu

TOL = 1e-4
ZERO_GRAD = ("linears.0.bias", "linears.6.bias", "linears.10.bias")


GRAD_TOL_GOLDEN = 2e-5     # gradients / BatchNorm buffers vs the reference's own numbers, relative to max|ref|
# Absolute floor under the relative bounds: a tensor whose TRUE value is zero holds only rounding
# noise on both sides (fixture tiny_u1_k5 has B = 2: a train-mode BatchNorm over two samples outputs
# +-1 whatever its input, so every gradient in front of it is exactly zero and the reference's own
# numbers there are ~1e-9 noise).  Real gradient tensors of the fixtures peak at 1e-3 ... 1e-2, six
# orders above it.
GRAD_ABS_FLOOR = 2e-9
GRAD_TOL_ORACLE = 5e-5     # the same vs the fp64 numpy oracle on random cases


def _close(a, b, tol=TOL, what=""):
    """Absolute bound (logits, losses, predictions: north_star's "within 1e-4 fp32")."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.isfinite(a).all(), what + ": non-finite values"
    err = np.abs(a - b).max() if a.size else 0.0
    scale = max(1.0, np.abs(b).max() if b.size else 1.0)
    record_margin("abs " + what, err / scale, tol)
    assert err <= tol * scale, "%s: max|d|=%.3e (scale %.3g)" % (what, err, scale)


def _close_rel(a, b, tol=GRAD_TOL_GOLDEN, what=""):
    """Relative to max|ref| of the tensor itself (gradients and buffers: a tensor whose entries
    are all of order 1e-3 must agree to tol of THAT, not of 1.0)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.isfinite(a).all(), what + ": non-finite values"
    err = np.abs(a - b).max() if a.size else 0.0
    scale = max(1e-12, np.abs(b).max() if b.size else 1.0)
    bound = tol * scale + GRAD_ABS_FLOOR
    record_margin("rel " + what, err / bound * tol, tol)
    assert err <= bound, "%s: max|d|=%.3e = %.3e of max|ref| %.3g (bound %.1e relative + %.0e)" % (
        what, err, err / scale, scale, tol, GRAD_ABS_FLOOR)


def _model(sd, U, k, L, T):
    from explainn_amd import ExplaiNN
    m = ExplaiNN(U, k, L, T)
    m.load_state_dict({key: torch.from_numpy(np.array(v)) for key, v in sd.items()})
    return m.cuda()


def _np(t):
    return t.detach().cpu().numpy()


def test_library_loaded_is_in_tree():
    from explainn_amd import _lib
    lib = _lib.load()
    assert _lib.LIB_PATH.endswith("explainn_amd/libexplainn_hip.so")
    assert lib.explainn_forward_train is not None


def test_eval_forward_golden(golden):
    g = golden
    m = _model(g.sd(), g.U, g.k, g.L, g.T).eval()
    x = torch.from_numpy(g.onehot()).cuda()
    with torch.no_grad():
        _close(_np(m(x)), g.z["eval/logits"], what="eval logits")
        xr = x.repeat(1, g.U, 1)
        outs = m.linears(xr)
        _close(_np(outs), g.z["eval/outs"], what="unit outputs")
        _close(_np(m.final(outs)), g.z["eval/logits"], what="final(outs)")
        if "eval/acts" in g.z.files:
            _close(_np(m.linears[:3](xr)), g.z["eval/acts"], what="activations")
        rev = torch.flip(x, dims=(1, 2))
        fwd_rev = np.stack([_np(m(x)), _np(m(rev))], axis=2)
        pred = np.concatenate([fwd_rev, fwd_rev.mean(2, keepdims=True),
                               fwd_rev.max(2, keepdims=True)], axis=2)
        _close(pred, g.z["eval/predict"], what="predict Fwd/Rev/Mean/Max")
    # the predict entry point (one chunked device pass, one transfer back) gives the same table
    from explainn_amd.predict import predict
    table = predict(m, g.onehot(), batch_size=3)
    assert table.dtype == np.float64 and table.shape == g.z["eval/predict"].shape
    _close(table, g.z["eval/predict"], what="predict()")
    assert np.array_equal(table, pred.astype(np.float64))


def _train_once(g, keep=None):
    m = _model(g.sd(), g.U, g.k, g.L, g.T).train()
    if keep is None:
        m.dropout_p = 0.0
    else:
        m.set_dropout_mask(torch.from_numpy(keep))
    x = torch.from_numpy(g.onehot()).cuda()
    y = torch.from_numpy(g.targets().astype(np.float32)).cuda()
    crit = torch.nn.BCEWithLogitsLoss() if g.loss_kind == "binary" else torch.nn.MSELoss()
    logits = m(x)
    loss = crit(logits, y)
    loss.backward()
    torch.cuda.synchronize()
    return m, logits, loss


def _golden_floor(g):
    """B = 2 fixtures: a train-mode BatchNorm over two samples outputs +-1/sqrt(1 + eps/var) whatever
    its input, so every gradient in front of one is an eps-sized cancellation residual (1e-5 of its
    summands) on both sides; those tensors are compared on an absolute 1e-7."""
    return 1e-7 if g.B <= 2 else GRAD_ABS_FLOOR


def test_train_forward_backward_golden(golden):
    g = golden
    m, logits, loss = _train_once(g)
    _close(_np(logits), g.z["train0/logits"], what="train logits")
    _close(loss.item(), g.z["train0/loss"], tol=1e-5, what="loss")
    params = dict(m.named_parameters())
    ref = g.group("train0/grad/")
    # knife-edge rows come from the oracle's intermediates at the fixture's parameters
    _, cache, _ = orc.forward(g.sd(), g.onehot(), training=True, return_cache=True)
    compare_grads([(k, _np(params[k].grad)) for k in ref], ref, GRAD_TOL_GOLDEN, cache, g.U, "golden ",
                  abs_floor=_golden_floor(g))
    if "train0/grad_rows/linears.6.weight" in g.z.files:
        # (the large fixture stores the first 200 rows = units 0 and 1 of this tensor only)
        from parity_util import knife_masks
        clean = ~knife_masks(cache, g.U)[0].reshape(-1)[:200]
        _close_rel(_np(params["linears.6.weight"].grad)[:200][clean],
                   g.z["train0/grad_rows/linears.6.weight"][clean], what="grad rows linears.6.weight")
    bufs = dict(m.named_buffers())
    for k, v in g.group("train0/buf/").items():
        if "tracked" in k:
            assert int(bufs[k].item()) == int(v), k
        else:
            _close_rel(_np(bufs[k]), v, what="buffer " + k)


def test_train_with_reference_dropout_mask(golden):
    g = golden
    m, logits, loss = _train_once(g, keep=g.keep_mask())
    _close(_np(logits), g.z["drop/logits"], what="dropout logits")
    _close(loss.item(), g.z["drop/loss"], tol=1e-5, what="loss")
    params = dict(m.named_parameters())
    ref = g.group("drop/grad/")
    _, cache, _ = orc.forward(g.sd(), g.onehot(), training=True, dropout_mask=g.keep_mask(), return_cache=True)
    compare_grads([(k, _np(params[k].grad)) for k in ref], ref, GRAD_TOL_GOLDEN, cache, g.U, "golden dropout ",
                  abs_floor=_golden_floor(g))


def test_adam_trajectory_golden(golden):
    """20 optimiser steps (torch.optim.Adam on our parameters, our gradients): per-step train-mode
    logits/loss and the learned filters stay within tolerance of the reference's trajectory.

    The gradient is discontinuous where a ReLU pre-activation crosses zero.  When some |y2| or |y3|
    is below 2e-6 at a step (found with the oracle at the model's own parameters), two fp32
    implementations may legitimately take different branches there and their trajectories part by
    a finite amount (seen: |y2| = 3.9e-7 at step 12 of small_u8_k19_mse; a single flip is 1/(100 B)
    of a unit's gradient, visible only in the small fixtures).  The run must follow the reference
    unless such a step has occurred; after leaving it, only single-step parity (logits vs the
    oracle at the same parameters) is asserted."""
    g = golden
    from explainn_amd import get_optimizer
    m = _model(g.sd(), g.U, g.k, g.L, g.T)
    m.dropout_p = 0.0
    opt = get_optimizer(m.parameters(), 0.003)
    crit = torch.nn.BCEWithLogitsLoss() if g.loss_kind == "binary" else torch.nn.MSELoss()
    n_steps = len(g.z["steps/loss"])
    if g.B <= 2:
        n_steps = 2           # see tests/test_oracle_golden.py
    knife_edge = None          # first step with a ReLU pre-activation within 2e-6 of zero
    following = True           # still on the reference's trajectory
    followed = 0
    for step in range(1, n_steps + 1):
        i = (step - 1) % g.n_batches
        xn = g.onehot(i)
        x = torch.from_numpy(xn).cuda()
        y = torch.from_numpy(g.targets(i).astype(np.float32)).cuda()
        sd_now = {k: _np(v) for k, v in m.state_dict().items()}
        ref_logits, cache, _ = orc.forward(sd_now, xn, training=True, return_cache=True)
        m.train()
        pred = m(x)
        loss = crit(pred, y)
        opt.zero_grad(); loss.backward(); opt.step()
        _close(_np(pred), ref_logits, what="logits vs oracle at own parameters, step %d" % step)
        if following:
            try:
                _close(loss.item(), g.z["steps/loss"][step - 1], tol=2e-5, what="loss step %d" % step)
                _close(_np(pred), g.z["steps/logits"][step - 1], what="logits step %d" % step)
                ref = g.group("step%d/sd/" % step)
                sd = m.state_dict()
                for k in ("linears.0.weight", "final.weight", "final.bias", "linears.11.weight"):
                    if k in ref:
                        _close(_np(sd[k]), ref[k], tol=2e-4, what="step %d %s" % (step, k))
                followed = step
            except AssertionError:
                # leaving the reference trajectory is only acceptable after a knife-edge step
                assert knife_edge is not None and knife_edge < step, \
                    "left the reference trajectory at step %d without a ReLU knife-edge" % step
                following = False
        if not following:
            # off the reference trajectory after a knife-edge: a flipped ReLU branch moves one
            # channel's gradient by one sample's share, and Adam moves a parameter by at most ~lr
            # per step whatever the gradient's size, so the runs can only part slowly
            since = step - knife_edge
            ref = g.group("step%d/sd/" % step)
            if "linears.0.weight" in ref:
                d = np.abs(_np(m.state_dict()["linears.0.weight"]) - ref["linears.0.weight"]).max()
                assert d <= 2 * 0.003 * since, "filters %.3g apart %d steps after the knife-edge" % (d, since)
            assert abs(loss.item() - g.z["steps/loss"][step - 1]) < 1e-2, step
        margin = min(np.abs(cache["y2"]).min(), np.abs(cache["y3"]).min())
        if margin < 2e-6 and knife_edge is None:
            knife_edge = step
    assert followed >= min(n_steps, 3), "trajectory left the reference after %d steps" % followed


@pytest.mark.parametrize("U,k,L,T,B,nfrac", [
    (5, 19, 61, 3, 24, 0.05),       # tail = 1, N bases, B < 64
    (7, 19, 200, 1, 130, 0.0),      # B not a multiple of 64, U not a multiple of 4
    (4, 7, 75, 2, 64, 0.1),
    (9, 26, 300, 4, 200, 0.01),     # n = 39 -> bucket 40 (zero-padded weights)
    (3, 19, 1000, 2, 70, 0.0),      # n = 140 (config C4's pooled length)
    (2, 19, 600, 5, 66, 0.02),      # n = 83  -> bucket 84 (config C5's pooled length)
    (37, 19, 61, 50, 70, 0.02),     # T = 50 (C3/C4): combiner forward/backward as MFMA GEMMs
    (70, 9, 40, 164, 131, 0.0),     # T = 164 (C5), ragged tiles in every GEMM dimension
    (3, 5, 33, 9, 5, 0.0),          # smallest GEMM case: one partly filled tile
    (6, 32, 120, 1, 40, 0.03),      # largest instantiated kernel size (two code words per window)
    (5, 2, 40, 2, 33, 0.05),        # smallest kernel size
    (4, 31, 260, 1, 20, 0.02),      # odd kernel size next to the maximum, two staging chunks (n = 32+)
    # large-n kernels over SEVERAL batch chunks (QCH / ACH > 1, ragged last chunk): qmom_big,
    # mid_big, passB<140>/<84>, fc_fwd<NQ > 32> -- the code paths configs C4 / C5 run
    (3, 19, 1000, 2, 300, 0.01),    # n = 140, 3 chunks of 128 (last one 44 sequences)
    (2, 19, 600, 5, 700, 0.02),     # n = 83, 6 chunks (last one 60)
    (3, 19, 450, 1, 90, 0.0),       # n = 61 -> bucket 64: two 32-wide k-steps of the bf16 fc_fwd
    (4, 19, 61, 2, 600, 0.0),       # few tasks, batch > 512: the per-unit head backward kernel (smaller
                                    # batches run it inside passA)
    (1100, 5, 40, 2, 70, 0.0),      # more units than threads in the combiner block that finishes BatchNorm3
    # the lower edge of a pooled-length bucket (n = previous bucket + 1): passA and qmom decide at
    # compile time which rows always / never exist inside the bucket (csrc/common.h: nq_lower)
    (3, 19, 109, 1, 70, 0.0),       # n = 13 -> bucket 16
    (3, 19, 165, 1, 70, 0.02),      # n = 21 -> bucket 24
    (3, 19, 186, 2, 70, 0.0),       # n = 24 = bucket 24's upper edge
    (3, 19, 193, 1, 70, 0.0),       # n = 25 -> bucket 26 (C2's kernels, one row short)
    (3, 19, 207, 1, 70, 0.0),       # n = 27 -> bucket 28
    (3, 19, 249, 1, 70, 0.0),       # n = 33 -> bucket 40 (first size with two row groups in passA)
    (3, 19, 305, 1, 70, 0.0),       # n = 41 -> bucket 48
    (2, 19, 529, 1, 70, 0.0),       # n = 73 -> bucket 84: the first n of the large-n kernels
    # the filter-bank GEMM's edges: one pooling window (a wave's first window is its last), the
    # widest kernel with one window, unit counts that leave a 32-unit tile / a two-tile group partly empty
    (3, 19, 25, 1, 9, 0.0),         # n = 1, Lo = 7
    (2, 32, 38, 1, 7, 0.0),         # k = 32, n = 1
    (33, 19, 32, 1, 31, 0.0),       # 33 units: tile 1 holds one unit
    (65, 4, 200, 1, 100, 0.02),     # 65 units: three tiles, the second group half empty; k = 4 is one k-step
])
def test_train_step_vs_oracle(U, k, L, T, B, nfrac):
    sd = orc.random_state_dict(U, k, L, T, seed=U + L)
    sd["linears.1.weight"][::2] *= -1
    x = orc.random_onehot(B, L, seed=4, n_frac=nfrac)
    rng = np.random.default_rng(5)
    y = (rng.random((B, T)) > 0.5).astype(np.float32)
    keep = (rng.random((B, 100 * U)) > 0.3).astype(np.uint8)
    ref_logits, _, ref_grads, nb = _oracle_step(sd, x, y, keep=keep)
    m = _model(sd, U, k, L, T).train()
    m.set_dropout_mask(torch.from_numpy(keep))
    logits = m(torch.from_numpy(x).cuda())
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.from_numpy(y).cuda())
    loss.backward()
    _close(_np(logits), ref_logits, what="logits")
    params = dict(m.named_parameters())
    _check_grads([(key, params[key].grad) for key in ref_grads], ref_grads)
    bufs = dict(m.named_buffers())
    for key, v in nb.items():
        if "tracked" not in key:
            _close_rel(_np(bufs[key]), v, tol=GRAD_TOL_ORACLE, what=key)
    # eval mode from the updated buffers
    m.eval()
    sd2 = dict(sd); sd2.update(nb)
    with torch.no_grad():
        _close(_np(m(torch.from_numpy(x).cuda())), orc.forward(sd2, x), what="eval logits")


_ORACLE_CACHES = {}       # id(reference gradient dict) -> (oracle cache, U): the knife-edge masks of that step


def _oracle_step(sd, x, y, freeze=0, keep=None, kind="binary"):
    """Logits, loss and BatchNorm buffers from the fp32 oracle (compared with absolute bounds);
    gradients from the oracle in FP64 -- the truth -- together with the error the reference's own
    fp32 arithmetic makes on this case (parity_util.reference_fp32_error), which sets the bar."""
    ref_logits, _, nb = orc.forward(sd, x, training=True, dropout_mask=keep, return_cache=True)
    lg64, cache, _ = orc.forward(sd, x, training=True, dropout_mask=keep, return_cache=True, dtype=np.float64)
    loss_fn = orc.bce_with_logits if kind == "binary" else orc.mse
    ref_loss, _ = loss_fn(ref_logits, y)
    _, dl = loss_fn(lg64, y.astype(np.float64))
    grads = orc.backward(cache, dl, freeze_top_n_filters=freeze)
    ref_err = reference_fp32_error(sd, x, y, grads, kind, keep, freeze=freeze, cache=cache)
    _ORACLE_CACHES[id(grads)] = (cache, sd["linears.0.weight"].shape[0], ref_err, grads)
    return ref_logits, ref_loss, grads, nb


def _check_grads(named, ref_grads, what=""):
    cache, U, ref_err, _ = _ORACLE_CACHES[id(ref_grads)]
    compare_grads([(key, _np(got)) for key, got in named], ref_grads, GRAD_TOL_ORACLE, cache, U, what, ref_err)


@pytest.mark.parametrize("freeze", [1, 3, 5])
def test_freeze_top_n_filters_vs_oracle(freeze):
    """The freeze hook (selene/__init__.py:254-257, 509-515: rows [0:n) of the filter gradient are
    zeroed) as the backward kernel applies it, against `oracle.backward(freeze_top_n_filters=n)`:
    through explainn_backward (autograd path) and through explainn_train_step (StepEngine)."""
    from explainn_amd.engine import StepEngine
    U, k, L, T, B = 5, 19, 61, 2, 40
    sd = orc.random_state_dict(U, k, L, T, seed=41)
    x = orc.random_onehot(B, L, seed=42, n_frac=0.02)
    y = (np.random.default_rng(43).random((B, T)) > 0.5).astype(np.float32)
    ref_logits, ref_loss, ref_grads, _ = _oracle_step(sd, x, y, freeze=freeze)
    assert np.abs(ref_grads["linears.0.weight"][:freeze]).max() == 0
    assert freeze == U or np.abs(ref_grads["linears.0.weight"][freeze:]).max() > 0
    xt, yt = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    # (a) explainn_forward_train + torch's loss + explainn_backward(freeze)
    m = _model(sd, U, k, L, T).train()
    m.dropout_p = 0.0
    m.freeze_top_n_filters = freeze
    logits = m(xt)
    torch.nn.functional.binary_cross_entropy_with_logits(logits, yt).backward()
    _close(_np(logits), ref_logits, what="logits")
    _check_grads([(key, p.grad) for key, p in m.named_parameters()], ref_grads, "autograd ")
    got = _np(m.linears[0].weight.grad)
    assert np.abs(got[:freeze]).max() == 0, "frozen rows must be exactly zero"
    # (b) explainn_train_step(freeze)
    m2 = _model(sd, U, k, L, T).train()
    m2.dropout_p = 0.0
    eng = StepEngine(m2, B, loss="binary")
    logits2, loss2 = eng.step(xt, yt, freeze_top_n_filters=freeze)
    torch.cuda.synchronize()
    _close(loss2.item(), ref_loss, tol=1e-5, what="loss")
    _check_grads([(key, v) for (key, _), v in zip(m2.named_parameters(), eng.views)], ref_grads, "step ")
    assert np.abs(_np(eng.views[0])[:freeze]).max() == 0
    assert torch.allclose(eng.views[0], m.linears[0].weight.grad, rtol=1e-4, atol=1e-7)


def test_large_context_small_batch_vs_oracle():
    """A context created for max_batch = 1024 (8 q-moment chunks, 8 passA chunks) used at B = 300:
    the trailing chunks are empty and must contribute exact zeros."""
    from explainn_amd.engine import StepEngine
    U, k, L, T, B = 3, 19, 200, 2, 300
    sd = orc.random_state_dict(U, k, L, T, seed=51)
    x = orc.random_onehot(B, L, seed=52, n_frac=0.01)
    y = (np.random.default_rng(53).random((B, T)) > 0.5).astype(np.float32)
    ref_logits, ref_loss, ref_grads, nb = _oracle_step(sd, x, y)
    m = _model(sd, U, k, L, T).train()
    m.dropout_p = 0.0
    eng = StepEngine(m, 1024, loss="binary")
    assert eng.ctx.max_batch == 1024
    logits, loss = eng.step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    torch.cuda.synchronize()
    _close(_np(logits), ref_logits, what="logits")
    _close(loss.item(), ref_loss, tol=1e-5, what="loss")
    _check_grads([(key, v) for (key, _), v in zip(m.named_parameters(), eng.views)], ref_grads)
    bufs = dict(m.named_buffers())
    for key, v in nb.items():
        if "tracked" not in key:
            _close_rel(_np(bufs[key]), v, tol=GRAD_TOL_ORACLE, what=key)


def test_eval_forward_invalidates_a_pending_backward():
    """`out = model(x)` in train mode, an eval-mode forward of the same batch size (a monitoring
    pass), then `loss.backward()`: the eval pass has overwritten the scratch the backward reads, so
    the backward must raise -- the reference's autograd graph would still hold its own
    activations; silently returning the eval batch's gradients is not an option."""
    import ctypes
    from explainn_amd import _lib
    sd = orc.random_state_dict(3, 5, 40, 1, seed=61)
    m = _model(sd, 3, 5, 40, 1).train()
    m.dropout_p = 0.0
    x = torch.from_numpy(orc.random_onehot(8, 40, seed=62)).cuda()
    out = m(x)
    m.eval()
    with torch.no_grad():
        m(x)                                           # same B: the batch-size check alone would pass
    m.train()
    with pytest.raises(RuntimeError, match="stale forward"):
        out.sum().backward()
    # the same at the C ABI: forward_train, forward_eval, backward -> E_STATE
    dev = m._device()
    ctx = m._context(8, dev)
    ps, keep = m._params_struct(dev)
    logits = torch.empty(8, 1, device=dev)
    assert ctx.lib.explainn_forward_train(ctx.handle, x.data_ptr(), 8, ctypes.byref(ps), None, 0.0,
                                          ctypes.c_uint64(0), logits.data_ptr(), None) == _lib.OK
    assert ctx.lib.explainn_forward_eval(ctx.handle, x.data_ptr(), 8, ctypes.byref(ps),
                                         logits.data_ptr(), None) == _lib.OK
    flat = torch.zeros(sum(p.numel() for p in m.parameters()), device=dev)
    gs = _lib.Grads()
    off = 0
    for field, p in zip(_lib.GRAD_FIELDS, m.parameters()):
        setattr(gs, field, flat[off:].data_ptr()); off += p.numel()
    dl = torch.ones(8, 1, device=dev)
    rc = ctx.lib.explainn_backward(ctx.handle, dl.data_ptr(), 8, ctypes.byref(ps), ctypes.byref(gs),
                                   0, None)
    torch.cuda.synchronize()
    assert rc == _lib.E_STATE


def test_batch_of_one_in_train_mode_raises():
    sd = orc.random_state_dict(2, 5, 26, 1)
    m = _model(sd, 2, 5, 26, 1).train()
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        m(torch.from_numpy(orc.random_onehot(1, 26)).cuda())


def test_soft_input_takes_the_dense_path():
    """The reference's forward accepts any float (B,4,L) tensor (architectures/__init__.py:111).  A
    batch that is not one-hot is detected before anything is computed from it and runs through
    csrc/dense.hip: eval logits, unit outputs, per-position activations, and a full train step
    (logits, loss, all 14 gradients, BatchNorm buffers) against the oracle on a soft input (a
    position-probability blend, one all-zero sequence, one exactly one-hot sequence)."""
    U, k, L, T, B = 5, 9, 47, 2, 20
    sd = orc.random_state_dict(U, k, L, T, seed=81)
    sd["linears.1.weight"][::2] *= -1
    rng = np.random.default_rng(82)
    x = rng.dirichlet(np.ones(4) * 0.3, size=(B, L)).transpose(0, 2, 1).astype(np.float32)
    x[0] = 0
    x[1] = orc.random_onehot(1, L, seed=83)[0]
    x = np.ascontiguousarray(x)
    xt = torch.from_numpy(x).cuda()
    m = _model(sd, U, k, L, T).eval()
    with torch.no_grad():
        _close(_np(m(xt)), orc.forward(sd, x), what="soft eval logits")
        _close(_np(m.linears(xt.repeat(1, U, 1))), orc.unit_outputs(sd, x), what="soft unit outputs")
        _close(_np(m.linears[:3](xt.repeat(1, U, 1))), orc.unit_activations(sd, x), what="soft activations")
        # a one-hot batch right after takes the fast path again, with the same context
        xo = orc.random_onehot(B, L, seed=84, n_frac=0.02)
        _close(_np(m(torch.from_numpy(xo).cuda())), orc.forward(sd, xo), what="one-hot after soft")
    y = (rng.random((B, T)) > 0.5).astype(np.float32)
    ref_logits, ref_loss, ref_grads, nb = _oracle_step(sd, x, y)
    m = _model(sd, U, k, L, T).train()
    m.dropout_p = 0.0
    logits = m(xt)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, torch.from_numpy(y).cuda())
    loss.backward()
    _close(_np(logits), ref_logits, what="soft train logits")
    _close(loss.item(), ref_loss, tol=1e-5, what="loss")
    _check_grads([(key, p.grad) for key, p in m.named_parameters()], ref_grads, "soft ")
    bufs = dict(m.named_buffers())
    for key, v in nb.items():
        if "tracked" not in key:
            _close_rel(_np(bufs[key]), v, tol=GRAD_TOL_ORACLE, what=key)
    # dense_input=True sends a ONE-HOT batch down the dense kernels too: same numbers as the fast path
    m2 = _model(sd, U, k, L, T).eval()
    m2.dense_input = True
    with torch.no_grad():
        _close(_np(m2(torch.from_numpy(xo).cuda())), orc.forward(sd, xo), what="one-hot via dense kernels")
    # dense_input=False keeps the strict behaviour
    m3 = _model(sd, U, k, L, T).eval()
    m3.dense_input = False
    with pytest.raises(ValueError, match="not one-hot"):
        m3(xt)
    m3(torch.from_numpy(xo).cuda())


def test_step_engine_takes_soft_input_like_forward():
    """The fused step (StepEngine, what Trainer uses) follows the model's validation schedule: a
    soft fp32 batch on the first steps is routed to the dense kernels, as forward() does (the
    reference accepts any float input, architectures/__init__.py:111), and dense_input=True goes
    there without validation."""
    from explainn_amd.engine import StepEngine
    U, k, L, T, B = 4, 9, 60, 2, 33
    sd = orc.random_state_dict(U, k, L, T, seed=61)
    rng = np.random.default_rng(62)
    x = rng.dirichlet(np.ones(4) * 0.4, size=(B, L)).transpose(0, 2, 1).astype(np.float32)
    x = np.ascontiguousarray(x)
    y = (rng.random((B, T)) > 0.5).astype(np.float32)
    ref_logits, ref_loss, ref_grads, _ = _oracle_step(sd, x, y)
    for dense in (None, True):
        m = _model(sd, U, k, L, T).train()
        m.dropout_p = 0.0
        m.dense_input = dense
        eng = StepEngine(m, B, loss="binary")
        for _ in range(3):                      # steps 1-2 validate, step 3 runs on the remembered route
            m.load_state_dict({key: torch.from_numpy(np.array(v)) for key, v in sd.items()})
            logits, loss = eng.step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
            _close(_np(logits), ref_logits, what="soft step logits (dense_input=%s)" % dense)
            _close(loss.item(), ref_loss, tol=1e-5, what="loss")
            _check_grads(zip([n for n, _ in m.named_parameters()], eng.views), ref_grads, "soft step ")


def _keep_mask_of_last_forward(m, B):
    """(U, B, 100) bool: the kept-bits words of the train forward in flight (explainn_debug_keep_bits)."""
    import ctypes as C
    U = m._options["cnn_units"]
    ctx = m._rt.ctx
    words = torch.empty(U, B, 4, dtype=torch.int32, device="cuda")
    from explainn_amd import _lib
    _lib.check(ctx.lib.explainn_debug_keep_bits(
        ctx.handle, B, words.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    w = words.cpu().numpy().view(np.uint32)
    bits = np.unpackbits(w.view(np.uint8).reshape(U, B, 16), axis=2, bitorder="little")
    return bits[:, :, :100].astype(bool)


def _corr(a, b):
    a = a.astype(np.float64).ravel() - a.mean(); b = b.astype(np.float64).ravel() - b.mean()
    return float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))


def test_builtin_dropout_generator_statistics():
    """The built-in generator is what bench.py times and Trainer trains with (nn.Dropout(0.3),
    architectures/__init__.py:92).  With BatchNorm2's weight 0 and bias 1 every pre-activation is
    1 > 0, so the stored "relu' > 0 and kept" bits ARE the keep mask.  Measured over 3 seeds at
    U = 32, B = 1024 (3.3 M draws each):
      * keep rate 0.7 (exactly 1 - 19661/65536) overall, per channel, per sequence, per unit;
      * no pairwise correlation between neighbouring channels (incl. the two 16-bit halves of one
        generator word and consecutive generator states), neighbouring sequences, sequences 16 and
        64 apart (the kernel's lane / tile strides), neighbouring units, and consecutive seeds;
      * scaling: replaying the extracted mask through the keep_mask path -- the path the
        reference's own masks pin, 1/(1-p) scaling included -- gives the same logits.
    Every bound is k sigma of the estimator with k chosen for the number of simultaneous tests
    (family-wise false-alarm rate < 1e-3)."""
    U, k, L, T, B = 32, 19, 200, 1, 1024
    sd = orc.random_state_dict(U, k, L, T, seed=5, perturb=False)
    sd["linears.7.weight"] = np.zeros_like(sd["linears.7.weight"])
    sd["linears.7.bias"] = np.ones_like(sd["linears.7.bias"])
    x = torch.from_numpy(orc.random_onehot(B, L, seed=6)).cuda()
    m = _model(sd, U, k, L, T).train()
    p_keep = 1.0 - 19661.0 / 65536.0
    var = p_keep * (1 - p_keep)
    masks, logits = [], []
    torch.manual_seed(1234)
    with torch.no_grad():
        for _ in range(3):
            logits.append(m(x).clone())
            masks.append(_keep_mask_of_last_forward(m, B))
    assert not np.array_equal(masks[0], masks[1]) and not np.array_equal(masks[1], masks[2])

    def rate_check(what, values, n_per, k_sigma):
        z = np.abs(values - p_keep).max() / np.sqrt(var / n_per)
        record_margin("dropout keep rate, " + what, z, k_sigma)
        assert z < k_sigma, "%s: keep rate off by %.2f sigma (bound %.1f)" % (what, z, k_sigma)

    def corr_check(what, c, n_samples, k_sigma):
        z = abs(c) * np.sqrt(n_samples)
        record_margin("dropout correlation, " + what, z, k_sigma)
        assert z < k_sigma, "%s: correlation %.2e = %.2f sigma (bound %.1f)" % (what, c, z, k_sigma)

    for s_i, M in enumerate(masks):
        tag = "seed %d " % s_i
        rate_check(tag + "overall", np.array([M.mean()]), M.size, 4.0)
        rate_check(tag + "per channel", M.mean(axis=(0, 1)), U * B, 4.6)
        rate_check(tag + "per sequence", M.mean(axis=(0, 2)), U * 100, 5.0)
        rate_check(tag + "per unit", M.mean(axis=(1, 2)), B * 100, 4.4)
        for r in range(99):                                   # neighbouring channels, pair by pair
            corr_check(tag + "channels r,r+1", _corr(M[:, :, r], M[:, :, r + 1]), U * B, 4.8)
        for d in (1, 2, 4):
            corr_check(tag + "channels r,r+%d pooled" % d, _corr(M[:, :, :-d], M[:, :, d:]), U * B * (100 - d), 4.2)
        for d in (1, 16, 64):
            corr_check(tag + "sequences b,b+%d" % d, _corr(M[:, :-d, :], M[:, d:, :]), U * (B - d) * 100, 4.2)
        corr_check(tag + "units u,u+1", _corr(M[:-1], M[1:]), (U - 1) * B * 100, 4.2)
        # three-way parity of one lane's consecutive draws (an xorshift stream is GF(2)-linear)
        par = M[:, :, 0:96:3] ^ M[:, :, 1:96:3] ^ M[:, :, 2:96:3]
        p3 = 0.5 * (1 - (1 - 2 * p_keep) ** 3)
        z = abs(par.mean() - p3) / np.sqrt(p3 * (1 - p3) / par.size)
        record_margin("dropout parity of three consecutive draws", z, 4.2)
        assert z < 4.2, z
    for a, b_ in ((0, 1), (1, 2)):
        corr_check("consecutive seeds", _corr(masks[a], masks[b_]), masks[a].size, 4.2)
    # scaling: the generator path and the keep-mask path (pinned to the reference by the drop/*
    # golden vectors) must be one computation
    keep = np.ascontiguousarray(np.transpose(masks[2], (1, 0, 2)).reshape(B, 100 * U)).astype(np.uint8)
    m.set_dropout_mask(torch.from_numpy(keep))
    with torch.no_grad():
        replay = m(x)
    assert torch.equal(replay, logits[2]), float((replay - logits[2]).abs().max())
    # same torch seed -> same masks; the accessor refuses when no train forward is in flight
    torch.manual_seed(1234)
    with torch.no_grad():
        again = m(x)
    assert torch.equal(again, logits[0])
    assert np.array_equal(_keep_mask_of_last_forward(m, B), masks[0])
    m.eval()
    with torch.no_grad():
        m(x)
    with pytest.raises(Exception, match="train-mode forward"):
        _keep_mask_of_last_forward(m, B)


@pytest.mark.parametrize("T,kind", [(1, "binary"), (3, "linear"), (6, "binary"), (50, "binary"),
                                    (21, "linear")])
def test_step_engine_vs_oracle(T, kind):
    """explainn_train_step (what bench.py times): loss value, logits and the flat gradient buffer
    against the oracle; T <= 4 takes the loss-fused head backward, T > 4 the separate loss kernel,
    T > 8 the MFMA GEMM head (and, at T = 50 with B = 100, more than one loss block would need
    N > 8192: covered by the C3-shaped property test)."""
    from explainn_amd.engine import StepEngine
    U, k, L, B = 6, 19, 200, 100
    sd = orc.random_state_dict(U, k, L, T, seed=11)
    x = orc.random_onehot(B, L, seed=12, n_frac=0.01)
    rng = np.random.default_rng(13)
    y = ((rng.random((B, T)) > 0.5) if kind == "binary" else rng.standard_normal((B, T))).astype(np.float32)
    m = _model(sd, U, k, L, T).train()
    m.dropout_p = 0.0
    eng = StepEngine(m, B, loss=kind)
    logits, loss = eng.step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    torch.cuda.synchronize()
    ref_logits, ref_loss, ref_grads, _ = _oracle_step(sd, x, y, kind=kind)
    _close(_np(logits), ref_logits, what="logits")
    _close(loss.item(), ref_loss, tol=1e-5, what="loss")
    _check_grads([(key, v) for (key, _), v in zip(m.named_parameters(), eng.views)], ref_grads)


# ---- base-code input: same numbers as the fp32 one-hot, bit for bit ------------
def test_base_codes_equal_onehot_path():
    from explainn_amd.architectures import BaseCodes
    from explainn_amd.engine import StepEngine
    from explainn_amd import predict as pr
    from explainn_amd import sequence as sq
    g = Golden("mid_u8_k19_L200")
    codes = g.codes[:g.B]
    x = torch.from_numpy(g.onehot()).cuda()
    c = torch.from_numpy(codes).cuda()
    m = _model(g.sd(), g.U, g.k, g.L, g.T).eval()
    with torch.no_grad():
        assert torch.equal(m(c), m(x))
        assert torch.equal(m(BaseCodes(c, True)), m(torch.flip(x, dims=(1, 2))))
        assert torch.equal(m.linears(c), m.linears(x.repeat(1, g.U, 1)))
        assert torch.equal(m.linears[:3](BaseCodes(c)), m.linears[:3](x))
    assert np.array_equal(sq.rc_codes(codes), sq.encode_codes_many(
        [sq.rc(s) for s in sq.one_hot_decode_many(g.onehot())]))
    assert np.array_equal(sq.codes_to_one_hot(codes), g.onehot())
    assert np.array_equal(pr.predict(m, codes, batch_size=5), pr.predict(m, g.onehot(), batch_size=5))
    # train step: identical logits, loss and flat gradient from either input form
    y = torch.from_numpy(g.targets().astype(np.float32)).cuda()
    res = []
    for inp in (x, c):
        mm = _model(g.sd(), g.U, g.k, g.L, g.T).train()
        eng = StepEngine(mm, g.B, loss=g.loss_kind)
        logits, loss = eng.step(inp, y, seed=7)
        res.append((logits.clone(), loss.clone(), eng.flat_grad.clone()))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    # autograd path
    mm = _model(g.sd(), g.U, g.k, g.L, g.T).train()
    mm.dropout_p = 0.0
    out = mm(BaseCodes(c))
    out.sum().backward()
    assert mm.final.weight.grad is not None and torch.isfinite(mm.final.weight.grad).all()


def test_base_codes_errors():
    from explainn_amd import _lib
    g = Golden("tiny_u3_k5_N")
    m = _model(g.sd(), g.U, g.k, g.L, g.T).eval()
    bad = torch.from_numpy(g.codes[:4].copy()).cuda()
    bad[0, 0] = 9                                              # not a base code
    with pytest.raises(ValueError):
        m(bad)
    with pytest.raises(RuntimeError):
        m(torch.zeros(4, g.L + 1, dtype=torch.uint8).cuda())    # wrong length
    # x == NULL without a staged batch is a state error, not a crash
    m2 = _model(g.sd(), g.U, g.k, g.L, g.T).eval()
    ctx = m2._context(4, m2._device())
    ps, keep = m2._params_struct(m2._device())
    out = torch.empty(4, g.T).cuda()
    import ctypes
    rc = ctx.lib.explainn_forward_eval(ctx.handle, None, 4, ctypes.byref(ps), out.data_ptr(), None)
    assert rc == _lib.E_STATE
    # ... and so is a staged batch of another size
    m2(torch.from_numpy(g.onehot()[:3]).cuda())
    rc = ctx.lib.explainn_forward_eval(ctx.handle, None, 4, ctypes.byref(ps), out.data_ptr(), None)
    assert rc == _lib.E_STATE


# ---- fused Adam (architectures/__init__.py:463-464 -> torch.optim.Adam defaults) ----------------
def test_fused_adam_matches_torch_adam():
    from explainn_amd import get_optimizer
    from explainn_amd.optim import FusedAdam
    torch.manual_seed(3)
    shapes = [(300, 4, 19), (300,), (30000, 26, 1), (1, 300), (1,), (5000,), (7, 3)]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa, ob = get_optimizer(pa, lr=0.003), torch.optim.Adam(pb, lr=0.003)
    assert isinstance(oa, FusedAdam) and isinstance(oa, torch.optim.Adam)
    for it in range(25):
        for a, b in zip(pa, pb):
            g = torch.randn_like(a) * (10.0 ** ((it % 5) - 3))
            a.grad, b.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), (a - b).abs().max()
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa["param_groups"][0].keys() == sb["param_groups"][0].keys()
    for i in sb["state"]:
        assert sa["state"][i].keys() == sb["state"][i].keys()
        assert float(sa["state"][i]["step"]) == float(sb["state"][i]["step"]) == 25.0
        for k in ("exp_avg", "exp_avg_sq"):
            ref = sb["state"][i][k]
            assert torch.allclose(sa["state"][i][k], ref, rtol=2e-6, atol=1e-6 * float(ref.abs().max()))
    # a state dict written by torch's Adam resumes in the fused one (checkpoint compatibility)
    oc = get_optimizer(pa, lr=0.003)
    oc.load_state_dict(sb)
    for a in pa:
        a.grad = torch.ones_like(a)
    oc.step()
    assert float(oc.state_dict()["state"][0]["step"]) == 26.0
    # options the kernel does not implement fall through to torch's own step
    od = FusedAdam(pa, lr=0.003, weight_decay=0.1)
    od.step()


# ---- the step in two halves, and the asynchronous RCCL path it exists for ------------------------
def test_split_step_equals_fused_step_and_overlapped_allreduce():
    import socket
    import torch.distributed as dist
    from explainn_amd.engine import StepEngine
    from explainn_amd.parallel import GradAllReduce
    g = Golden("small_u8_k19")
    x = torch.from_numpy(g.onehot()).cuda()
    y = torch.from_numpy(g.targets().astype(np.float32)).cuda()
    ref = None
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        for mode in ("fused", "split"):
            m = _model(g.sd(), g.U, g.k, g.L, g.T).train()
            eng = StepEngine(m, g.B, loss=g.loss_kind)
            assert eng.conv_grad_elements == g.U * 4 * g.k + 3 * g.U
            sync = None
            if mode == "split":
                # one-rank group: the collectives are identities, but they are real asynchronous
                # RCCL launches on RCCL's stream, ordered against the step by events
                sync = GradAllReduce(eng.flat_grad, split=eng.conv_grad_elements, force=True)
            for it in range(3):
                logits, loss = eng.step(x, y, seed=11 + it, freeze_top_n_filters=2, grad_sync=sync)
            torch.cuda.synchronize()
            out = (logits.clone(), loss.clone(), eng.flat_grad.clone())
            if ref is None:
                ref = out
            else:
                for a, b in zip(ref, out):
                    assert torch.equal(a, b)
    finally:
        dist.destroy_process_group()
    # state check of the C ABI: the conv half needs its fc half
    m = _model(g.sd(), g.U, g.k, g.L, g.T).train()
    eng = StepEngine(m, g.B, loss=g.loss_kind)
    from explainn_amd import _lib
    import ctypes
    rc = eng.ctx.lib.explainn_train_step_conv(eng.ctx.handle, g.B, ctypes.byref(eng.ps),
                                              ctypes.byref(eng.gs), 0, None)
    assert rc == _lib.E_STATE


def test_smaller_batch_after_larger_one_on_the_same_context():
    """A ragged batch (70) after a larger one (128) on the same context: the lanes of the last,
    partly filled 64-sequence tile still hold the larger batch's intermediates and must not leak
    into the gradients (the last epoch batch of every Trainer run looks like this)."""
    from explainn_amd.engine import StepEngine
    U, k, L, T = 6, 19, 200, 2
    sd = orc.random_state_dict(U, k, L, T, seed=31)
    m = _model(sd, U, k, L, T).train()
    m.dropout_p = 0.0
    eng = StepEngine(m, 128, loss="binary")
    rng = np.random.default_rng(32)
    xb = orc.random_onehot(128, L, seed=33, n_frac=0.01)
    yb = (rng.random((128, T)) > 0.5).astype(np.float32)
    eng.step(torch.from_numpy(xb).cuda(), torch.from_numpy(yb).cuda())
    m.load_state_dict({key: torch.from_numpy(np.array(v)) for key, v in sd.items()})   # undo BN buffer updates
    xs, ys = xb[:70] * 1.0, yb[:70]
    xs = orc.random_onehot(70, L, seed=34, n_frac=0.01)
    logits, loss = eng.step(torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda())
    torch.cuda.synchronize()
    ref_logits, ref_loss, ref_grads, _ = _oracle_step(sd, xs, ys)
    _close(_np(logits), ref_logits, what="logits")
    _close(loss.item(), ref_loss, tol=1e-5, what="loss")
    _check_grads([(key, v) for (key, _), v in zip(m.named_parameters(), eng.views)], ref_grads)


def test_eval_tables_follow_parameter_changes():
    """Inside an eval_cache() scope the eval entry points keep their folded tables (filter LUTs,
    BatchNorm folds, FC1 fragments) between calls and rebuild them when explainn_params.version
    moves.  Every way the values can change that torch can see must move it: an in-place torch
    update, a train-mode step of this package (BatchNorm buffers written through raw pointers),
    the fused Adam launch, load_state_dict, a re-assigned Parameter; a write through `.data`
    (selene/__init__.py:294 `final.weight.data.clamp_(0)`), which torch's version counters do NOT
    see, needs invalidate() inside a scope -- and nothing at all outside one, where every call
    rebuilds the tables."""
    from explainn_amd import get_optimizer
    U, k, L, T, B = 4, 9, 60, 2, 24
    sd = orc.random_state_dict(U, k, L, T, seed=71)
    m = _model(sd, U, k, L, T)
    x = orc.random_onehot(B, L, seed=72, n_frac=0.02)
    xt = torch.from_numpy(x).cuda()

    def check(what, scoped=True):
        cur = {key: _np(v) for key, v in m.state_dict().items()}
        m.eval()
        with torch.no_grad():
            a = m(xt)
            b = m(xt)                                 # second call: cached tables inside a scope
        assert torch.equal(a, b)
        _close(_np(a), orc.forward(cur, x), what=what)

    with m.eval_cache():
        check("initial")
        with torch.no_grad():
            m.linears[0].weight.mul_(1.5)                 # in-place torch op
            m.linears[7].running_var.add_(0.3)
        check("after in-place updates")
        m.train()
        m.dropout_p = 0.0
        out = m(xt)                                       # train forward: buffers move, tables overwritten
        out.sum().backward()
        check("after a train-mode forward/backward")
        opt = get_optimizer(m.parameters(), 0.01)
        m.train()
        torch.nn.functional.binary_cross_entropy_with_logits(m(xt), torch.ones(B, T).cuda()).backward()
        opt.step()                                        # fused Adam: raw-pointer update
        check("after a fused Adam step")
        m.load_state_dict({key: torch.from_numpy(np.array(v)) for key, v in sd.items()})
        check("after load_state_dict")
        m.final.weight = torch.nn.Parameter(m.final.weight.detach() * 2)
        check("after re-assigning a Parameter")
        # a `.data` write inside the scope: invisible to torch's counters, so the owner of the scope
        # says so
        m.final.weight.data.clamp_(0)
        m.linears[0].weight.data[1] += 0.25
        m.invalidate()
        check("after .data writes + invalidate()")
    # outside a scope nothing is cached: `.data` writes need no announcement
    check("outside a scope")
    m.final.weight.data.mul_(-1.0)
    m.linears[0].weight.data[0] -= 0.5
    m.linears[7].running_mean.data.add_(0.1)
    check("after .data writes outside a scope")
    # nested scopes and a scope opened after writes made outside it
    m.linears[1].weight.data.mul_(1.1)
    with m.eval_cache():
        check("scope opened after a .data write")
        with m.eval_cache():
            check("nested scope")


def test_input_validation_schedule():
    """validate_input=True: the first two forwards read the validation flag before computing (a
    soft batch is routed to the dense kernels and the model keeps validating per call from then
    on); later forwards of a model that has only seen one-hot input enqueue without a host sync and
    the sticky flag is read by check_input() / every 64th call: a soft batch then raises instead of
    passing silently.  'always' routes every batch."""
    U, k, L, T, B = 3, 9, 60, 1, 16
    sd = orc.random_state_dict(U, k, L, T, seed=91)
    xo = orc.random_onehot(B, L, seed=92)
    rng = np.random.default_rng(93)
    xs = rng.random((B, 4, L)).astype(np.float32)
    xs /= xs.sum(1, keepdims=True)
    xo_t, xs_t = torch.from_numpy(xo).cuda(), torch.from_numpy(xs).cuda()
    m = _model(sd, U, k, L, T).eval()
    with torch.no_grad():
        for _ in range(4):
            _close(_np(m(xo_t)), orc.forward(sd, xo), what="one-hot, deferred validation")
        m.check_input()                               # nothing flagged so far
        m(xs_t)                                       # call 5: not validated before computing ...
        with pytest.raises(ValueError, match="not one-hot"):
            m.check_input()                           # ... but reported here
        m.check_input()                               # the flag was cleared by the read
        m.validate_input = "always"
        _close(_np(m(xs_t)), orc.forward(sd, xs), what="soft batch under validate_input='always'")
        _close(_np(m(xo_t)), orc.forward(sd, xo), what="one-hot batch after it")
    m2 = _model(sd, U, k, L, T).eval()
    with torch.no_grad():
        _close(_np(m2(xs_t)), orc.forward(sd, xs), what="soft batch on the first call")
        for _ in range(3):                            # a model that met soft input keeps routing per call
            _close(_np(m2(xs_t)), orc.forward(sd, xs), what="soft batch on later calls")
            _close(_np(m2(xo_t)), orc.forward(sd, xo), what="one-hot between soft batches")


def test_fc_bf16_piece_products_are_fp32_accurate():
    """fc_fwd runs on the bf16 matrix core with BOTH fp32 operands split exactly into three bf16
    pieces and all nine piece products accumulated in fp32.  Claim under test: that is
    an fp32 computation, not a reduced-precision one -- against the fp64 oracle the unit outputs
    are at fp32 rounding level (round 2 measured 3.3e-7 of scale for this form and 4.1e-7 for the
    fp32-MFMA form of the same contraction; a bf16-input GEMM would be off by ~1e-3)."""
    U, k, L, T, B = 24, 19, 200, 1, 256
    sd = orc.random_state_dict(U, k, L, T, seed=77)
    x = orc.random_onehot(B, L, seed=78)
    ref = orc.unit_outputs(sd, x, dtype=np.float64)
    m = _model(sd, U, k, L, T).eval()
    xt = torch.from_numpy(x).cuda()
    with torch.no_grad():
        bf = _np(m.linears(xt.repeat(1, U, 1)))
    scale = max(1.0, np.abs(ref).max())
    e_bf = np.abs(bf - ref).max() / scale
    record_margin("abs unit outputs vs fp64 (bf16 nine-piece fc_fwd)", e_bf, 3e-6)
    assert e_bf <= 3e-6, e_bf


def test_eval_replica_shares_tensors_and_overlaps_on_a_second_stream():
    """ExplaiNN.eval_replica(): a second handle on the same parameter tensors with its own device
    context.  Two batches in flight on two streams give the numbers of two forwards in a row, bit for
    bit; a parameter change through the model is seen by the replica."""
    from explainn_amd.architectures import BaseCodes
    U, k, L, T = 9, 19, 200, 3
    sd = orc.random_state_dict(U, k, L, T, seed=5)
    m = _model(sd, U, k, L, T).eval()
    rep = m.eval_replica()
    assert rep is m.eval_replica() and rep is not m
    assert all(a is b for a, b in zip(m.parameters(), rep.parameters()))
    rng = np.random.default_rng(1)
    c1 = torch.from_numpy(rng.integers(0, 4, size=(300, L)).astype(np.uint8)).cuda()
    c2 = torch.from_numpy(rng.integers(0, 4, size=(300, L)).astype(np.uint8)).cuda()
    with torch.no_grad():
        want1, want2 = m(c1).clone(), m(BaseCodes(c2, reverse_complement=True)).clone()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.no_grad(), m.eval_cache(), rep.eval_cache():
        for _ in range(20):
            with torch.cuda.stream(s1):
                got1 = m(c1)
            with torch.cuda.stream(s2):
                got2 = rep(BaseCodes(c2, reverse_complement=True))
    torch.cuda.synchronize()
    assert torch.equal(got1, want1) and torch.equal(got2, want2)
    with torch.no_grad():
        m.final.bias.add_(1.0)
        assert torch.allclose(rep(c1), want1 + 1.0, atol=1e-6)

