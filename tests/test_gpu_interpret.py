"""GPU: the device-side filter -> PWM export (csrc/interpret.hip through the C ABI and
explainn_amd/interpret.py) against the reference-fed fixtures, and -- at the C2 shape, streamed in
several batches -- against a dense numpy recount of the product's own float16 activations."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from test_interpret_oracle import PFM_CASES, load, onehot  # noqa: E402
from oracle import interpret_oracle as io  # noqa: E402

pytestmark = pytest.mark.gpu


def _model(z, m):
    from explainn_amd import ExplaiNN
    net = ExplaiNN(m["U"], m["k"], m["L"], m["T"])
    net.load_state_dict({k[3:]: torch.from_numpy(np.array(z[k])) for k in z.files if k.startswith("sd/")})
    return net.cuda().eval()


@pytest.mark.parametrize("name", PFM_CASES)
@pytest.mark.parametrize("batch", [1024, 7])
def test_filter_pwms_golden(name, batch):
    from explainn_amd import interpret as it
    z, m = load(name)
    net = _model(z, m)
    x = onehot(z["codes"])
    outs, preds = it._get_outs_preds(net, x, batch_size=16)
    for mine, ref in ((outs, z["outs"]), (preds, z["preds"])):
        ulp = np.maximum(np.abs(ref.astype(np.float64)), 2.0 ** -14) * 2.0 ** -10
        assert (np.abs(mine.astype(np.float64) - ref.astype(np.float64)) <= ulp).all()
    idxs = z["idxs"]
    res = it.filter_pwms(net, x, idxs, m["rc"], batch_size=batch, site_cap=m["cap"])
    assert res["thresholds"].dtype == np.float16
    assert np.array_equal(res["thresholds"], z["thresholds"]), (res["thresholds"], z["thresholds"])
    assert np.array_equal(res["nsites"], z["nsites"]), (res["nsites"], z["nsites"])
    assert np.array_equal(res["pfm"], z["pfm"])
    hit_ref = (z["acts"] > z["thresholds"][None, :, None]).any(axis=2)
    sel = np.zeros(len(x), dtype=bool)
    sel[idxs] = True
    if m["rc"]:
        sel[idxs + len(x) // 2] = True
    assert np.array_equal(res["hit"][sel], hit_ref[sel])
    assert not res["hit"][~sel].any()
    for u, (s_u, imps) in enumerate(it.filter_importances(z["outs"], z["sd/final.weight"], idxs, res["hit"])):
        assert np.array_equal(s_u, z["imp_sel/%d" % u])
        assert np.array_equal(imps, z["imp/%d" % u])


def test_dense_export_matches_reference_float16():
    from explainn_amd import interpret as it
    from torch.utils.data import DataLoader, TensorDataset
    z, m = load("pfm_u8_k9")
    net = _model(z, m)
    x = torch.from_numpy(onehot(z["codes"]))
    loader = DataLoader(TensorDataset(x, torch.zeros(len(x), 1)), batch_size=20)
    acts, outs, preds = it._get_acts_outs_preds(net, loader)
    assert acts.dtype == np.float16 and acts.shape == z["acts"].shape
    assert (acts == z["acts"]).mean() > 0.995
    ulp = np.maximum(np.abs(z["acts"].astype(np.float64)), 2.0 ** -14) * 2.0 ** -10
    assert (np.abs(acts.astype(np.float64) - z["acts"].astype(np.float64)) <= ulp).all()


def _dense_recount(acts16, codes, sel_idx, k, cap):
    """numpy recount from a dense float16 activation array, first `cap` sites per unit in
    (sequence, position) order."""
    N, U, Lo = acts16.shape
    thr = 0.5 * np.amax(acts16[sel_idx], axis=(0, 2))
    pfm = np.zeros((U, k, 4), dtype=np.int64)
    nsites = np.zeros(U, dtype=np.int64)
    for u in range(U):
        ii, jj = np.where(acts16[sel_idx, u, :] > thr[u])
        ii, jj = ii[:cap], jj[:cap]
        nsites[u] = len(ii)
        if len(ii) == 0:
            continue
        sites = codes[sel_idx[ii][:, None], jj[:, None] + np.arange(k)[None, :]]     # (S,k)
        for a in range(4):
            pfm[u, :, a] = (sites == a).sum(axis=0)
    return thr, pfm, nsites


@pytest.mark.parametrize("cap", [io.SITE_CAP, 300])
def test_c2_shape_streamed_against_dense_recount(cap):
    """300 units, 200 bp, 700 sequences in batches of 256 (ragged tail), 1 % N, every third
    sequence selected.  The device export must equal a recount from the dense float16 activations
    the same model produces through model.linears[:3] (the reference's own data flow)."""
    from explainn_amd import ExplaiNN, interpret as it
    torch.manual_seed(5)
    U, k, L, N = 300, 19, 200, 700
    net = ExplaiNN(U, k, L, 1).cuda().eval()
    g = np.random.default_rng(9)
    codes = g.integers(0, 4, size=(N, L)).astype(np.uint8)
    codes[g.random((N, L)) < 0.01] = 4
    x = onehot(codes)
    idxs = np.arange(0, N, 3)
    acts = np.zeros((N, U, L - k + 1), dtype=np.float16)
    with torch.no_grad():
        for i in range(0, N, 100):
            acts[i:i + 100] = net.linears[:3](torch.from_numpy(x[i:i + 100]).cuda()).cpu().numpy()
    thr, pfm, nsites = _dense_recount(acts, codes, idxs, k, cap)
    res = it.filter_pwms(net, x, idxs, False, batch_size=256, site_cap=cap)
    assert np.array_equal(res["thresholds"], thr)
    assert np.array_equal(res["nsites"], nsites)
    assert np.array_equal(res["pfm"], pfm)
    if cap < io.SITE_CAP:
        assert nsites.max() == cap


def test_export_needs_eval_mode_and_checks_shapes():
    from explainn_amd import ExplaiNN
    net = ExplaiNN(4, 5, 30, 1).cuda()
    x = torch.zeros(3, 4, 30).cuda()
    x[:, 0, :] = 1
    with pytest.raises(NotImplementedError):
        net.train().filter_act_max(x, torch.zeros(4).cuda())
    net.eval()
    with pytest.raises(RuntimeError):
        net.filter_sites(x, torch.zeros(3).cuda(), torch.zeros(4, dtype=torch.int32).cuda(),
                         torch.zeros(4, 5, 4, dtype=torch.int32).cuda())


@pytest.mark.parametrize("scoring", ["max", "sum"])
def test_pwm_module_golden(scoring):
    """explainn_amd.PWM against the reference PWM module's scores (tolerance 1e-4 relative to the
    largest score: fp32 sums in a different order)."""
    import os
    from conftest import GOLDEN
    from explainn_amd import PWM
    z = np.load(os.path.join(GOLDEN, "pwm_scan.npz"), allow_pickle=False)
    mod = PWM(z["pwms"], z["x"].shape[2], scoring)
    assert sorted(mod.state_dict().keys()) == [str(s) for s in z["state_keys"]]
    assert not any(p.requires_grad for p in mod.parameters())
    with pytest.raises(RuntimeError):
        mod(torch.from_numpy(z["x"]))                        # no CPU fallback
    got = mod.cuda()(torch.from_numpy(z["x"]).cuda()).cpu().numpy()
    assert got.shape == z[scoring].shape
    assert np.abs(got - z[scoring]).max() <= 1e-4 * max(1.0, np.abs(z[scoring]).max())


def test_pwm_scan_wide_bank_vs_oracle():
    """A 300-matrix bank over 200 bp (ragged last quad, k = 19) against the fp64 oracle."""
    from explainn_amd import PWM
    from oracle import explainn_oracle as eo
    g = np.random.default_rng(3)
    pwms = g.standard_normal((301, 4, 19)).astype(np.float32)
    codes = g.integers(0, 5, size=(33, 200)).astype(np.uint8)
    x = onehot(codes)
    for scoring in ("max", "sum"):
        got = PWM(pwms, 200, scoring).cuda()(torch.from_numpy(x).cuda()).cpu().numpy()
        ref = eo.pwm_scan(pwms, x, scoring)
        assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())


def test_interpret_entry_point_writes_reference_layout(tmp_path):
    """explainn_amd.interpret.main on a saved checkpoint + TSV: the files interpret.py:128-235 leaves
    behind (output-layer weights, filter importances, one JASPAR motif per filter), with the counts
    the fixture holds."""
    import gzip
    from explainn_amd import interpret as it
    from explainn_amd import sequence as sq
    z, m = load("pfm_u8_k9")
    net = _model(z, m)
    seqs = sq.one_hot_decode_many(onehot(z["codes"]))
    tsv = tmp_path / "train.tsv"
    with open(tsv, "wt") as fh:
        for i, (s, y) in enumerate(zip(seqs, z["labels"])):
            fh.write("seq%d\t%s\t%s\n" % (i, s, "\t".join(str(int(v)) for v in y)))
    ckpt = tmp_path / "best_model.pth.tar"
    torch.save({"step": 1, "arch": "ExplaiNN", "options": dict(net._options),
                "state_dict": {k: v.cpu() for k, v in net.state_dict().items()}, "min_loss": 0.0,
                "optimizer": {}}, ckpt)
    out = tmp_path / "out"
    it.main([str(ckpt), str(tsv), "-n", "demo", "-o", str(out), "-b", "16", "-t"])
    weights = np.loadtxt(out / "output-layer-weights.tsv", skiprows=1, usecols=1, ndmin=1)
    assert np.allclose(weights, z["sd/final.weight"][0], atol=1e-6)
    for u in range(m["U"]):
        txt = open(out / "motifs" / ("filter%d.jaspar" % u)).read()
        if z["nsites"][u] == 0:
            assert txt == ""
            continue
        lines = txt.splitlines()
        assert lines[0] == ">filter%d demo" % u
        counts = np.array([[float(v) for v in ln[ln.index("[") + 1:ln.index("]")].split()] for ln in lines[1:]])
        assert np.array_equal(counts.T, z["pfm"][u])
    with gzip.open(out / "filter-importances.tsv.gz", "rt") as fh:
        rows = fh.read().splitlines()
    assert len(rows) - 1 == sum(len(z["imp_sel/%d" % u]) for u in range(m["U"]))
    assert (out / "time-interpret.py.txt").exists()
