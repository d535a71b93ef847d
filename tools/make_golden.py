#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the IMPORTED reference (read-only, /root/reference).

Runs only in the build container (the reference never travels to the GPU box); the .npz
files it writes are data: inputs (base codes, targets, the reference's own randomly
initialised state dict) and the outputs the reference computed for them.

    python tools/make_golden.py            # rewrites every fixture

`Bio` (biopython) is not installed and is not used by the ExplaiNN class; the three stub
modules below only satisfy the reference's top-of-file imports.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
for name in ("Bio", "Bio.motifs", "Bio.Seq"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["Bio"].motifs = sys.modules["Bio.motifs"]
sys.modules["Bio.Seq"].Seq = object
sys.path.insert(0, "/root/reference/explainn")

import numpy as np
import torch

from architectures import ExplaiNN, PWM, get_loss, get_optimizer   # noqa: E402  (reference)
import sequence as ref_sequence                                 # noqa: E402  (reference)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
torch.set_num_threads(1)


def make_codes(B, L, seed, n_frac):
    g = torch.Generator().manual_seed(seed)
    codes = torch.randint(0, 4, (B, L), generator=g)
    if n_frac > 0:
        holes = torch.rand(B, L, generator=g) < n_frac
        codes[holes] = 4
    return codes.numpy().astype(np.uint8)


def codes_to_onehot(codes):
    B, L = codes.shape
    x = np.zeros((B, 4, L), dtype=np.float32)
    for a in range(4):
        x[:, a, :] = (codes == a)
    return x


def perturb_bn(model, seed, negative_gamma):
    """Move BN affine params and running stats off their defaults so eval mode is a real test."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for i in (1, 7, 11):
            bn = model.linears[i]
            bn.weight.copy_(1 + 0.3 * torch.randn(bn.weight.shape, generator=g))
            bn.bias.copy_(0.2 * torch.randn(bn.bias.shape, generator=g))
            bn.running_mean.copy_(0.1 * torch.randn(bn.bias.shape, generator=g))
            bn.running_var.copy_(0.5 + torch.rand(bn.bias.shape, generator=g))
        if negative_gamma:
            model.linears[1].weight[::2] *= -1      # sign-aware pooling must handle gamma1 < 0


def sd_np(model, prefix="sd/"):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def grads_np(model, prefix):
    return {prefix + k: p.grad.detach().numpy().copy() for k, p in model.named_parameters()}


def make_case(name, U, k, L, T, B, seed, n_frac=0.0, negative_gamma=False, loss_kind="binary",
              n_steps=20, n_batches=3, keep_full=True, tandem=False):
    torch.manual_seed(seed)
    model = ExplaiNN(U, k, L, T)
    perturb_bn(model, seed + 1, negative_gamma)
    codes = make_codes(B * n_batches, L, seed + 2, n_frac)
    if tandem:
        # tandem repeats: equal activations inside pooling windows (tie -> first index)
        unit = np.array([0, 1, 2, 3, 0, 1, 2], dtype=np.uint8)
        codes[0] = np.resize(unit, L)
        codes[1] = 0
    x_all = torch.from_numpy(codes_to_onehot(codes))
    g = torch.Generator().manual_seed(seed + 3)
    if loss_kind == "binary":
        y_all = (torch.rand(B * n_batches, T, generator=g) > 0.5).float()
    else:
        y_all = torch.randn(B * n_batches, T, generator=g)
    out = dict(cfg=np.array([U, k, L, T, B, n_batches], dtype=np.int64),
               loss_kind=np.array(loss_kind), codes=codes, y=y_all.numpy())
    out.update(sd_np(model, "sd/"))
    x, y = x_all[:B], y_all[:B]
    crit = get_loss("binary" if loss_kind == "binary" else "linear")

    # ---- eval mode: logits, unit outputs, per-position activations (test.py:148-160) ----
    model.eval()
    with torch.no_grad():
        out["eval/logits"] = model(x).numpy()
        xr = x.repeat(1, U, 1)
        out["eval/outs"] = model.linears(xr).numpy()
        if keep_full:
            out["eval/acts"] = model.linears[:3](xr).numpy()
        # predict.py:75-94
        rev = torch.from_numpy(np.ascontiguousarray(
            ref_sequence.rc_one_hot_encoding_many(x.numpy())))
        f_ = model(x).numpy()[:, :, None]; r_ = model(rev).numpy()[:, :, None]
        fr = np.concatenate((f_, r_), axis=2)
        out["eval/predict"] = np.concatenate(
            (f_, r_, fr.mean(axis=2, keepdims=True), fr.max(axis=2, keepdims=True)), axis=2
        ).astype(np.float64)

    # ---- train mode, dropout p=0: one fwd+bwd (selene/__init__.py:283-291) ----
    state0 = {k_: v.clone() for k_, v in model.state_dict().items()}
    model.train()
    model.linears[9].p = 0.0
    model.zero_grad()
    logits = model(x)
    loss = crit(logits, y)
    loss.backward()
    out["train0/logits"] = logits.detach().numpy()
    out["train0/loss"] = np.array(loss.item())
    gr = grads_np(model, "train0/grad/")
    if not keep_full:
        big = gr.pop("train0/grad/linears.6.weight")
        rng = np.random.default_rng(7)
        out["train0/grad_proj/linears.6.weight"] = np.array(
            [(big[:, :, 0] * rng.standard_normal(big.shape[:2])).sum(), np.abs(big).sum()])
        out["train0/grad_rows/linears.6.weight"] = big[:200].copy()
    out.update(gr)
    out.update({"train0/buf/" + k_: v.numpy().copy() for k_, v in model.state_dict().items()
                if "running" in k_ or "tracked" in k_})

    # ---- train mode with the reference's own dropout; keep-mask captured by a hook ----
    model.load_state_dict(state0)
    model.linears[9].p = 0.3
    captured = {}
    hook = model.linears[9].register_forward_hook(
        lambda mod, inp, outp: captured.__setitem__("keep", (outp != 0).squeeze(-1)))
    torch.manual_seed(seed + 4)
    model.zero_grad()
    logits = model(x)
    loss = crit(logits, y)
    loss.backward()
    hook.remove()
    # where the ReLU output is 0 the mask is irrelevant; (out != 0) is a valid keep-mask
    out["drop/keep_bits"] = np.packbits(captured["keep"].numpy().astype(np.uint8), axis=1)
    out["drop/logits"] = logits.detach().numpy()
    out["drop/loss"] = np.array(loss.item())
    for key in ("linears.0.weight", "linears.1.weight", "linears.1.bias", "linears.7.weight",
                "linears.10.weight", "linears.11.weight", "final.weight", "final.bias"):
        out["drop/grad/" + key] = dict(model.named_parameters())[key].grad.numpy().copy()

    # ---- n_steps Adam steps (lr 0.003), p=0, batches cycled ----
    model.load_state_dict(state0)
    model.linears[9].p = 0.0
    opt = get_optimizer(model.parameters(), 0.003)
    losses, lg = [], []
    for step in range(1, n_steps + 1):
        i = (step - 1) % n_batches
        xb, yb = x_all[i * B:(i + 1) * B], y_all[i * B:(i + 1) * B]
        model.train()
        pred = model(xb)
        ls = crit(pred, yb)
        opt.zero_grad(); ls.backward(); opt.step()
        losses.append(ls.item()); lg.append(pred.detach().numpy().copy())
        if step in (1, 5, 20):
            snap = sd_np(model, "step%d/sd/" % step)
            if not keep_full:
                snap = {k_: v for k_, v in snap.items() if "linears.6" not in k_
                        and "linears.7" not in k_}
            out.update(snap)
            if step == 1:
                st = opt.state_dict()["state"]
                names = [k_ for k_, _ in model.named_parameters()]
                for j, nm in enumerate(names):
                    if keep_full or "linears.6" not in nm and "linears.7" not in nm:
                        out["step1/adam/exp_avg/" + nm] = st[j]["exp_avg"].numpy().copy()
                        out["step1/adam/exp_avg_sq/" + nm] = st[j]["exp_avg_sq"].numpy().copy()
    out["steps/loss"] = np.array(losses)
    out["steps/logits"] = np.stack(lg)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.1f KB" % (name, os.path.getsize(path) / 1024))


def make_trainer_case():
    """A short reference `Trainer.train_and_validate()` run (selene/__init__.py:248-271) on a tiny
    synthetic data set: dropout off, loaders unshuffled -> deterministic; the values it logs to
    train.txt / validation.txt and the checkpoint's bookkeeping are the expected outputs."""
    import tempfile
    from torch.utils.data import DataLoader, TensorDataset
    from architectures import get_metrics
    from selene import Trainer
    U, k, L, T, B, N = 6, 11, 60, 1, 16, 64
    torch.manual_seed(21)
    model = ExplaiNN(U, k, L, T)
    model.linears[9].p = 0.0
    codes = make_codes(N + 32, L, 22, 0.01)
    g = torch.Generator().manual_seed(23)
    # labels correlated with the presence of a motif so the metrics are meaningful
    motif = np.array([0, 1, 2, 3, 0, 1], dtype=np.uint8)
    y = np.zeros((N + 32, T), dtype=np.float32)
    for i in range(N + 32):
        if torch.rand(1, generator=g).item() < 0.5:
            pos = int(torch.randint(0, L - len(motif), (1,), generator=g).item())
            codes[i, pos:pos + len(motif)] = motif
            y[i, 0] = 1.0
    x = torch.from_numpy(codes_to_onehot(codes)); yt = torch.from_numpy(y)
    loaders = {"train": DataLoader(TensorDataset(x[:N], yt[:N]), B, shuffle=False),
               "validation": DataLoader(TensorDataset(x[N:], yt[N:]), B, shuffle=False)}
    out = dict(cfg=np.array([U, k, L, T, B, N], dtype=np.int64), codes=codes, y=y)
    out.update(sd_np(model, "sd/"))
    steps_per_epoch = N // B
    with tempfile.TemporaryDirectory() as d:
        tr = Trainer(model, loaders, get_loss("binary"), get_metrics("binary"),
                     get_optimizer(model.parameters(), 0.003), max_steps=steps_per_epoch * 3,
                     patience=steps_per_epoch * 10, report_stats_every_n_steps=steps_per_epoch,
                     output_dir=d, cpu_n_threads=1, use_cuda=False, logging_verbosity=0)
        tr.train_and_validate()
        train_txt = open(os.path.join(d, "train.txt")).read().split()
        val_lines = open(os.path.join(d, "validation.txt")).read().strip().split("\n")
        # written seconds ago by this very process (min_loss is a numpy scalar, which the
        # weights_only unpickler rejects); not a file that ships with the reference
        ck = torch.load(os.path.join(d, "best_model.pth.tar"), weights_only=False)
    out["train_txt"] = np.array([float(v) for v in train_txt[1:]])
    out["val_header"] = np.array(val_lines[0])
    out["val_txt"] = np.array([[float(v) for v in ln.split("\t")] for ln in val_lines[1:]])
    out["ck_step"] = np.array(ck["step"]); out["ck_min_loss"] = np.array(ck["min_loss"])
    out["ck_arch"] = np.array(ck["arch"])
    out["ck_keys"] = np.array(sorted(ck.keys()))
    out["ck_filters"] = ck["state_dict"]["linears.0.weight"].numpy()
    np.savez_compressed(os.path.join(OUT, "trainer_run.npz"), **out)
    print("trainer_run", out["train_txt"], out["val_txt"])


def make_transfer_case():
    """Transfer learning / filter freezing exactly as the reference drives it: the reference's own
    `train._train` (train.py:304-342) is imported and called with `filter_weights` (and `freeze`
    on / off).  Only two things are patched, both outside the code under test: click_option_group
    (absent here, used only by the CLI decorators) is stubbed, and the `ExplaiNN` name inside the
    reference's train module is wrapped so that the model it builds has Dropout p = 0 (the run must
    be deterministic) and so that the model object can be read back after training (`_train`
    returns nothing).  Loaders are unshuffled.

    What the reference does (and what the fixture pins): the optimiser is built BEFORE
    `linears[0].weight` is re-assigned to a new nn.Parameter (train.py:314 vs :324), so Adam keeps
    the stale Parameter, which never receives a gradient again -- the filter bank stays exactly at
    `filter_weights` whether `freeze` is set or not; every other parameter trains normally."""
    import tempfile
    import types as _types
    cog = _types.ModuleType("click_option_group")

    class _OptGroup:
        @staticmethod
        def group(*a, **k):
            return lambda f: f

        @staticmethod
        def option(*a, **k):
            return lambda f: f
    cog.optgroup = _OptGroup
    sys.modules["click_option_group"] = cog
    import train as ref_train                                   # the reference's train.py
    from torch.utils.data import DataLoader, TensorDataset
    U, k, L, T, B, N = 6, 9, 50, 1, 16, 64
    codes = make_codes(N + 32, L, 52, 0.01)
    g = torch.Generator().manual_seed(53)
    y = (torch.rand(N + 32, T, generator=g) > 0.5).float()
    x = torch.from_numpy(codes_to_onehot(codes))
    fw = [0.5 * torch.randn(4, k, generator=g) for _ in range(U + 2)]       # two more than needed
    out = dict(cfg=np.array([U, k, L, T, B, N], dtype=np.int64), codes=codes, y=y.numpy(),
               filter_weights=torch.stack(fw).numpy())
    steps_per_epoch = N // B
    for tag, freeze in (("plain", False), ("freeze", True)):
        loaders = {"train": DataLoader(TensorDataset(x[:N], y[:N]), B, shuffle=False),
                   "validation": DataLoader(TensorDataset(x[N:], y[N:]), B, shuffle=False)}
        built = {}

        def _explainn_p0(*a, **kw):
            m = ExplaiNN(*a, **kw)
            m.linears[9].p = 0.0
            built["model"] = m
            built["sd0"] = {k_: v.clone() for k_, v in m.state_dict().items()}
            return m
        ref_train.ExplaiNN = _explainn_p0
        torch.manual_seed(51)
        with tempfile.TemporaryDirectory() as d:
            ref_train._train(L, T, loaders, "binary", steps_per_epoch, cnn_units=U, kernel_size=k,
                             lr=0.003, max_epochs=2, patience=10, cpu_threads=1, output_dir=d,
                             filter_weights=fw, freeze=freeze)
            train_txt = open(os.path.join(d, "train.txt")).read().split()
            val_lines = open(os.path.join(d, "validation.txt")).read().strip().split("\n")
        m = built["model"]
        if tag == "plain":
            out.update({"sd/" + k_: v.numpy().copy() for k_, v in built["sd0"].items()})
        out[tag + "/train_txt"] = np.array([float(v) for v in train_txt[1:]])
        out[tag + "/val_loss"] = np.array([float(ln.split("\t")[0]) for ln in val_lines[1:]])
        for k_, v in m.state_dict().items():
            if "running" in k_ or "tracked" in k_:
                continue
            out[tag + "/final/" + k_] = v.detach().numpy().copy()
        moved = float((m.linears[0].weight.detach() - torch.stack(fw[:U])).abs().max())
        print("transfer/%s: filters moved %.3g, final.weight moved %.3g" % (
            tag, moved, float((m.final.weight.detach() - built["sd0"]["final.weight"]).abs().max())))
        # fewer filters than units is an IndexError in the reference (train.py:320)
    try:
        ref_train._train(L, T, loaders, "binary", steps_per_epoch, cnn_units=U, kernel_size=k,
                         max_epochs=1, output_dir=tempfile.mkdtemp(), filter_weights=fw[:U - 1])
        out["short_raises"] = np.array("none")
    except Exception as e:                                       # noqa: BLE001
        out["short_raises"] = np.array(type(e).__name__)
    import logging
    for nm in ("selene", "train", "validation"):
        logging.getLogger(nm).handlers.clear()
    np.savez_compressed(os.path.join(OUT, "transfer.npz"), **out)
    print("transfer", out["plain/train_txt"], out["freeze/train_txt"], out["short_raises"])


def make_encoding_case():
    seqs = ["ACGT", "acgtn", "NNACGTRYACGT", "TTTTGGGGCCCCAAAA", "A"]
    out = {}
    for i, s in enumerate(seqs):
        enc = ref_sequence.one_hot_encode(s)
        out["seq%d" % i] = np.array(s)
        out["enc%d" % i] = enc
        out["rc%d" % i] = np.ascontiguousarray(ref_sequence.rc_one_hot_encoding(enc))
    np.savez_compressed(os.path.join(OUT, "encoding.npz"), **out)
    print("encoding")


def make_pwm_case():
    """The reference `PWM` module (architectures/__init__.py:116-170) on random log-odds matrices:
    max and sum scoring, with N columns and one soft (non-one-hot) sequence."""
    g = np.random.default_rng(40)
    G, k, L, B = 6, 11, 50, 9
    pwms = g.standard_normal((G, 4, k)).astype(np.float32)
    codes = make_codes(B, L, 41, 0.05)
    x = codes_to_onehot(codes)
    x[-1] = g.random((4, L)).astype(np.float32)          # the module is a plain convolution
    out = {"pwms": pwms, "x": x}
    for scoring in ("max", "sum"):
        with torch.no_grad():
            out[scoring] = PWM(pwms, L, scoring)(torch.from_numpy(x)).numpy()
    out["state_keys"] = np.array(sorted(PWM(pwms, L).state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, "pwm_scan.npz"), **out)
    print("pwm_scan", out["max"].shape)


def make_pfm_case(name, U, k, L, T, N, seed, rev_complement, loss_kind="binary", n_frac=0.0, cap=None):
    """Filter -> PWM fixture.  The float16 activations / unit outputs / predictions
    come from the imported reference model exactly as test.py:128-166 extracts them (x.repeat ->
    model.linears, model.final, model.linears[:3]); the bookkeeping after that (well-predicted
    selection, thresholds, site counting, importances) has no importable form here -- interpret.py
    compiles a C tool into the reference tree at import time -- so it is oracle/interpret_oracle.py's
    restatement applied to those reference activations."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import interpret_oracle as io
    torch.manual_seed(seed)
    model = ExplaiNN(U, k, L, T)
    perturb_bn(model, seed + 1, False)
    model.eval()
    codes = make_codes(N, L, seed + 2, n_frac)
    # plant a motif in half of the sequences so that some filters see repeated strong sites
    g = np.random.default_rng(seed + 3)
    motif = g.integers(0, 4, size=k).astype(np.uint8)
    for i in range(0, N, 2):
        s0 = int(g.integers(0, L - k + 1))
        codes[i, s0:s0 + k] = motif
    x = codes_to_onehot(codes)
    if loss_kind == "binary":
        labels = (g.random((N, T)) > 0.5).astype(np.float64)
    else:
        labels = g.standard_normal((N, T))
    if rev_complement:
        x = np.append(x, ref_sequence.rc_one_hot_encoding_many(x), axis=0)
        labels = np.append(labels, labels, axis=0)
        rc_codes = (3 - codes[:, ::-1]).astype(np.int16)
        rc_codes[codes[:, ::-1] == 4] = 4
        codes = np.append(codes, rc_codes.astype(np.uint8), axis=0)
    Lo = L - k + 1
    acts = np.zeros((len(x), U, Lo), dtype=np.float16)
    outs = np.zeros((len(x), U), dtype=np.float16)
    preds = np.zeros((len(x), T), dtype=np.float16)
    with torch.no_grad():
        bs = 16
        for i0 in range(0, len(x), bs):
            Xs = torch.Tensor(x[i0:i0 + bs]).repeat(1, U, 1)
            o = model.linears(Xs)
            outs[i0:i0 + bs] = o.numpy()
            preds[i0:i0 + bs] = model.final(o).numpy()
            acts[i0:i0 + bs] = model.linears[:3](Xs).numpy()
    if loss_kind != "binary":
        # a regression target that the (random) model "predicts": prediction + noise, so that the
        # top-5% intersection of interpret.py:345-359 is not empty
        labels = preds.astype(np.float64) + 0.05 * g.standard_normal(preds.shape)
    # selection with the reference's own torch expression where one is used (interpret.py:326,330)
    idxs = io.well_predicted_sequences(preds, labels, loss_kind, rev_complement)
    if loss_kind == "binary" and not rev_complement:
        p = torch.sigmoid(torch.from_numpy(preds)).numpy()
        ref_idxs = np.where((labels == (p > .5).astype(int)).all(axis=1))[0]
        assert np.array_equal(idxs, ref_idxs)
    thr = io.act_thresholds(acts, idxs, rev_complement)
    kw = {} if cap is None else {"cap": cap}
    pfm, nsites = io.site_pfms(codes, acts, idxs, thr, k, rev_complement, **kw)
    imps = io.filter_importances(outs, model.final.weight.detach().numpy(), idxs, acts, thr)
    out = sd_np(model)
    out.update({"codes": codes, "labels": labels, "acts": acts, "outs": outs, "preds": preds,
                "idxs": idxs, "thresholds": thr, "pfm": pfm, "nsites": nsites,
                "meta": np.array([U, k, L, T, len(x), int(rev_complement),
                                  0 if loss_kind == "binary" else 1,
                                  io.SITE_CAP if cap is None else cap], dtype=np.int64)})
    for u, (sel, im) in enumerate(imps):
        out["imp_sel/%d" % u] = sel
        out["imp/%d" % u] = im
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "selected", len(idxs), "sites", nsites.tolist())


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    make_encoding_case()
    if len(sys.argv) > 1 and sys.argv[1] == "trainer":
        make_trainer_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "transfer":
        make_transfer_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pwm":
        make_pwm_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pfm":
        #             name              U   k   L  T   N seed  rc
        make_pfm_case("pfm_u8_k9",      8,  9, 60, 1, 48, 30, False)
        make_pfm_case("pfm_u8_k9_rc",   8,  9, 60, 2, 48, 31, True, n_frac=0.02)
        make_pfm_case("pfm_u6_k19_cap", 6, 19, 80, 1, 40, 32, False, cap=25)
        make_pfm_case("pfm_u5_k7_lin",  5,  7, 50, 1, 120, 33, False, loss_kind="linear")
        sys.exit(0)
    #          name               U   k   L   T   B  seed
    make_case("tiny_u1_k5",       1,  5,  26, 1,  2, 10)
    make_case("tiny_u3_k5_N",     3,  5,  26, 3, 16, 11, n_frac=0.1, negative_gamma=True)
    make_case("small_u8_k19",     8, 19,  60, 3, 16, 12, negative_gamma=True)
    make_case("small_u8_k19_mse", 8, 19,  61, 2, 16, 13, n_frac=0.02, loss_kind="linear")
    make_case("tandem_u3_k5",     3,  5,  40, 1,  8, 14, tandem=True)
    make_case("mid_u8_k19_L200",  8, 19, 200, 1, 16, 15, n_frac=0.01, negative_gamma=True)
    make_case("c1_u100_k19_L200", 100, 19, 200, 1, 64, 16, n_steps=5, n_batches=2,
              keep_full=False)
