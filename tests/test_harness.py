"""Host-side harness (Trainer, _train helpers, predict, sequence): CPU logic tests plus GPU runs
checked against the reference `Trainer` golden run (tests/golden/trainer_run.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def test_sequence_matches_reference_vectors():
    from explainn_amd import sequence as s
    z = np.load(GOLDEN + "/encoding.npz", allow_pickle=False)
    for i in range(5):
        q = str(z["seq%d" % i]); e = s.one_hot_encode(q)
        assert e.dtype == np.float64 and np.array_equal(e, z["enc%d" % i])
        assert np.array_equal(s.rc_one_hot_encoding(e), z["rc%d" % i])
    assert s.rc("AACGTN") == "NACGTT"
    assert s.one_hot_decode(s.one_hot_encode("ACNGT")) == "ACNGT"
    many = s.one_hot_encode_many(["ACGT", "TTNN"])
    assert many.shape == (2, 4, 4) and np.array_equal(s.rc_one_hot_encoding_many(many)[0],
                                                      many[0][::-1, ::-1])


def test_loader_avoids_single_sample_last_batch():
    from explainn_amd.train import _avoid_single_sample_batch, _get_data_loader
    assert _avoid_single_sample_batch(101, 100) == 99          # 101 % 100 == 1 -> shrink
    assert _avoid_single_sample_batch(100, 100) == 100
    assert _avoid_single_sample_batch(7, 3) == 1               # 7%3==1 -> 2, 7%2==1 -> 1
    dl = _get_data_loader(np.zeros((101, 4, 30)), np.zeros((101, 1)), 100)
    assert all(b[0].shape[0] > 1 for b in dl)


def test_tsv_reader_and_rc_augmentation(tmp_path):
    from explainn_amd.train import _get_seqs_labels_ids
    p = tmp_path / "d.tsv"
    p.write_text("a\tACGTAC\t1\t0.5\nb\tNNGTAC\t0\t1.5\n")
    seqs, labels, ids = _get_seqs_labels_ids(str(p), reverse_complement=True)
    assert seqs.shape == (4, 4, 6) and labels.shape == (4, 2) and list(ids) == ["a", "b", "a", "b"]
    assert np.array_equal(seqs[2], seqs[0][::-1, ::-1]) and seqs[1][:, 0].sum() == 0


def _trainer_fixture():
    z = np.load(os.path.join(GOLDEN, "trainer_run.npz"), allow_pickle=False)
    U, k, L, T, B, N = [int(v) for v in z["cfg"]]
    codes = z["codes"]
    x = np.zeros((codes.shape[0], 4, L), dtype=np.float32)
    for a in range(4):
        x[:, a, :] = codes == a
    sd = {key[3:]: torch.from_numpy(np.array(z[key])) for key in z.files if key.startswith("sd/")}
    return z, (U, k, L, T, B, N), torch.from_numpy(x), torch.from_numpy(z["y"]), sd


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False])
def test_trainer_run_matches_reference(tmp_path, fused):
    """3 epochs of Trainer.train_and_validate on the reference's data/initialisation: the values
    written to train.txt / validation.txt and the checkpoint bookkeeping match the reference run
    (dropout off; train-mode losses and learned filters within 1e-4)."""
    from torch.utils.data import DataLoader, TensorDataset
    from explainn_amd import ExplaiNN, get_loss, get_metrics, get_optimizer
    from explainn_amd.selene import Trainer, _load_checkpoint_file
    z, (U, k, L, T, B, N), x, y, sd = _trainer_fixture()
    model = ExplaiNN(U, k, L, T)
    model.load_state_dict(sd)
    model.dropout_p = 0.0
    loaders = {"train": DataLoader(TensorDataset(x[:N], y[:N]), B, shuffle=False),
               "validation": DataLoader(TensorDataset(x[N:], y[N:]), B, shuffle=False)}
    crit = get_loss("binary")
    if not fused:
        crit = torch.nn.BCEWithLogitsLoss(reduction="mean", pos_weight=torch.ones(1))  # autograd path
    spe = N // B
    tr = Trainer(model, loaders, crit, get_metrics("binary"),
                 get_optimizer(model.parameters(), 0.003), max_steps=spe * 3, patience=spe * 10,
                 report_stats_every_n_steps=spe, output_dir=str(tmp_path), cpu_n_threads=1,
                 use_cuda=True, logging_verbosity=0)
    assert tr._fused_step_available() == fused
    tr.train_and_validate()
    train_txt = open(tmp_path / "train.txt").read().split()
    assert train_txt[0] == "loss"
    got = np.array([float(v) for v in train_txt[1:]])
    assert np.abs(got - z["train_txt"]).max() < 1e-4, (got, z["train_txt"])
    lines = open(tmp_path / "validation.txt").read().strip().split("\n")
    assert lines[0] == str(z["val_header"])
    val = np.array([[float(v) for v in ln.split("\t")] for ln in lines[1:]])
    # eval-mode numbers after independent training carry the reference's noise-driven drift of the
    # three pre-BatchNorm biases (their true gradient is zero; Adam random-walks them on rounding
    # noise, and running_mean lags behind -- SURVEY.md 7.2), so validation loss is compared at
    # 5e-4 and the rank metrics of 32 near-chance validation points only loosely
    assert np.abs(val[:, 0] - z["val_txt"][:, 0]).max() < 5e-4
    assert np.abs(val[:, 1:] - z["val_txt"][:, 1:]).max() < 0.1
    ck = _load_checkpoint_file(str(tmp_path / "best_model.pth.tar"))
    assert sorted(ck.keys()) == list(z["ck_keys"]) and ck["arch"] == str(z["ck_arch"])
    assert ck["step"] == int(z["ck_step"]) and abs(ck["min_loss"] - float(z["ck_min_loss"])) < 5e-4
    assert np.abs(ck["state_dict"]["linears.0.weight"].numpy() - z["ck_filters"]).max() < 1e-4


@pytest.mark.gpu
def test_train_entry_point_and_predict_roundtrip(tmp_path):
    """`_train` (train.py:304) end to end on a toy problem, then `_load_model` + `predict` on the
    checkpoint it wrote: [Fwd, Rev, Mean, Max] consistent with direct eval-mode calls."""
    from torch.utils.data import DataLoader, TensorDataset
    from explainn_amd.predict import _load_model, predict
    from explainn_amd.train import _train
    z, (U, k, L, T, B, N), x, y, sd = _trainer_fixture()
    loaders = {"train": DataLoader(TensorDataset(x[:N], y[:N]), B, shuffle=True),
               "validation": DataLoader(TensorDataset(x[N:], y[N:]), B)}
    torch.manual_seed(0)
    _train(L, T, loaders, "binary", N // B, cnn_units=U, kernel_size=k, max_epochs=2, patience=5,
           output_dir=str(tmp_path))
    for name in ("train.txt", "validation.txt", "best_model.pth.tar", "selene.log"):
        assert (tmp_path / name).exists(), name
    model = _load_model(str(tmp_path / "best_model.pth.tar"))
    assert not model.training and model._options["cnn_units"] == U
    preds = predict(model, x[:10].numpy(), batch_size=4)
    assert preds.shape == (10, T, 4) and preds.dtype == np.float64
    with torch.no_grad():
        fwd = model(x[:10].cuda()).cpu().numpy()
        rev = model(torch.flip(x[:10], dims=(1, 2)).cuda()).cpu().numpy()
    assert np.abs(preds[:, :, 0] - fwd).max() < 1e-6 and np.abs(preds[:, :, 1] - rev).max() < 1e-6
    assert np.allclose(preds[:, :, 2], (fwd + rev) / 2, atol=1e-6)
    assert np.allclose(preds[:, :, 3], np.maximum(fwd, rev), atol=1e-6)


def test_constructor_contract_on_cpu():
    """The façade keeps the reference's constructor surface (architectures/__init__.py:44-107):
    _options keys, state_dict keys/shapes, and the same rejections torch's layers raise there."""
    import torch
    from explainn_amd import ExplaiNN
    m = ExplaiNN(3, 5, 40, 2)
    assert list(m._options) == ["cnn_units", "kernel_size", "sequence_length", "n_features", "weights_file"]
    sd = m.state_dict()
    n = (40 - 5 + 1) // 7
    assert sd["linears.0.weight"].shape == (3, 4, 5) and sd["linears.6.weight"].shape == (300, n, 1)
    assert sd["linears.10.weight"].shape == (3, 100, 1) and sd["final.weight"].shape == (2, 3)
    assert m.__class__.__name__ == "ExplaiNN"
    for bad in ((0, 5, 40, 1), (3, 5, 40, 0), (3, 0, 40, 1), (3, 19, 24, 1)):
        with pytest.raises(ValueError):
            ExplaiNN(*bad)
    with pytest.raises(RuntimeError):          # no CPU fallback: the model must be on a HIP device
        m(torch.zeros(2, 4, 40))


def test_codes_helpers_roundtrip():
    from explainn_amd import sequence as sq
    seqs = ["ACGTNACGT", "ttgacnnAC"]
    codes = sq.encode_codes_many(seqs)
    assert codes.dtype == np.uint8 and codes.shape == (2, 9)
    assert np.array_equal(sq.codes_to_one_hot(codes, float), sq.one_hot_encode_many(seqs))
    assert np.array_equal(sq.codes_to_one_hot(sq.rc_codes(codes), float),
                          sq.rc_one_hot_encoding_many(sq.one_hot_encode_many(seqs)))
    with pytest.raises(ValueError):
        sq.encode_codes_many(["ACG", "AC"])


def test_batched_loader_yields_what_the_stock_loader_yields():
    """train._get_data_loader cuts batches with one index op; order and contents must be those of
    the reference's DataLoader(TensorDataset(...), batch_size, shuffle) under the same RNG state."""
    import torch
    from torch.utils.data import DataLoader, TensorDataset
    from explainn_amd.train import _get_data_loader
    rng = np.random.default_rng(0)
    seqs = rng.random((37, 4, 11)).astype(np.float32)
    labels = rng.random((37, 2)).astype(np.float32)
    for shuffle in (False, True):
        torch.manual_seed(5)
        ours = _get_data_loader(seqs, labels, batch_size=12, shuffle=shuffle)
        assert isinstance(ours, DataLoader) and len(ours.dataset) == 37
        got = [b for _ in range(2) for b in ours]                       # two epochs
        torch.manual_seed(5)
        ref = DataLoader(TensorDataset(torch.Tensor(seqs), torch.Tensor(labels)), ours.batch_size,
                         shuffle=shuffle)
        want = [b for _ in range(2) for b in ref]
        assert len(got) == len(want) == 2 * len(ours)
        for (gx, gy), (wx, wy) in zip(got, want):
            assert torch.equal(gx, wx) and torch.equal(gy, wy)
    assert ours.batch_size == 11                  # 37 % 12 == 1: shrunk so that no batch has one sample


# ---- transfer learning / filter freezing (train.py:316-324, selene/__init__.py:254-257) ----------
def _transfer_fixture():
    z = np.load(os.path.join(GOLDEN, "transfer.npz"), allow_pickle=False)
    U, k, L, T, B, N = [int(v) for v in z["cfg"]]
    codes = z["codes"]
    x = np.zeros((codes.shape[0], 4, L), dtype=np.float32)
    for a in range(4):
        x[:, a, :] = codes == a
    sd = {key[3:]: torch.from_numpy(np.array(z[key])) for key in z.files if key.startswith("sd/")}
    return z, (U, k, L, T, B, N), torch.from_numpy(x), torch.from_numpy(z["y"]), sd


def test_transfer_learning_needs_a_filter_per_unit():
    """Fewer pre-trained filters than units: the reference indexes filter_weights[i] for every unit
    (train.py:320) and raises IndexError; so does the drop-in (before any device work)."""
    from explainn_amd.train import _train
    z, (U, k, L, T, B, N), x, y, sd = _transfer_fixture()
    assert str(z["short_raises"]) == "IndexError"
    fw = [torch.from_numpy(w) for w in z["filter_weights"][:U - 1]]
    with pytest.raises(IndexError):
        _train(L, T, {}, "binary", 4, cnn_units=U, kernel_size=k, filter_weights=fw)


@pytest.mark.gpu
@pytest.mark.parametrize("freeze", [False, True])
def test_transfer_learning_matches_reference(tmp_path, monkeypatch, freeze):
    """`_train(filter_weights=..., freeze=...)` against the run of the reference's own `_train`
    (tests/golden/transfer.npz): the reference builds Adam before it re-assigns
    `linears[0].weight`, so the given filters never move -- frozen or not -- while every other
    parameter trains; train.txt / validation loss / final parameters match."""
    from torch.utils.data import DataLoader, TensorDataset
    import explainn_amd.train as tr
    z, (U, k, L, T, B, N), x, y, sd = _transfer_fixture()
    tag = "freeze" if freeze else "plain"
    built = {}

    def factory(*a, **kw):
        m = tr_ExplaiNN(*a, **kw)
        m.load_state_dict(sd)
        m.dropout_p = 0.0
        built["model"] = m
        return m
    tr_ExplaiNN = tr.ExplaiNN
    monkeypatch.setattr(tr, "ExplaiNN", factory)
    loaders = {"train": DataLoader(TensorDataset(x[:N], y[:N]), B, shuffle=False),
               "validation": DataLoader(TensorDataset(x[N:], y[N:]), B, shuffle=False)}
    fw = [torch.from_numpy(w) for w in z["filter_weights"]]
    trainer = tr._train(L, T, loaders, "binary", N // B, cnn_units=U, kernel_size=k, lr=0.003,
                        max_epochs=2, patience=10, cpu_threads=1, output_dir=str(tmp_path),
                        filter_weights=fw, freeze=freeze)
    assert trainer.freeze_top_n_filters == (U if freeze else 0)
    m = built["model"]
    filters = m.linears[0].weight.detach().cpu().numpy()
    assert np.array_equal(filters, z["filter_weights"][:U]), "the filter bank must not move"
    assert np.array_equal(filters, z[tag + "/final/linears.0.weight"])
    train_txt = np.array([float(v) for v in open(tmp_path / "train.txt").read().split()[1:]])
    assert np.abs(train_txt - z[tag + "/train_txt"]).max() < 1e-4, (train_txt, z[tag + "/train_txt"])
    lines = open(tmp_path / "validation.txt").read().strip().split("\n")[1:]
    val = np.array([float(ln.split("\t")[0]) for ln in lines])
    # eval-mode numbers after independent training: the reference's Adam random-walks the three
    # pre-BatchNorm biases on rounding noise and running_mean lags behind; with
    # only 8 steps behind the buffers that is 7e-4 here
    assert np.abs(val - z[tag + "/val_loss"]).max() < 2e-3
    final = {key: v.detach().cpu().numpy() for key, v in m.state_dict().items()}
    for key in ("final.weight", "final.bias", "linears.1.weight", "linears.7.weight",
                "linears.10.weight", "linears.11.weight", "linears.11.bias", "linears.6.weight"):
        ref = z[tag + "/final/" + key]
        assert np.abs(final[key] - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), key
        assert np.abs(final[key] - sd[key].numpy()).max() > 1e-4, key + " must have trained"


@pytest.mark.gpu
def test_validation_batches_larger_than_train_batches(tmp_path):
    """101 training sequences at batch_size 100: train.py:297-302 shrinks only the TRAIN loader (to
    99), so validation batches (100) are larger than any train batch.  The eval forward then makes
    the model replace its device context by a larger one; the fused step must pick the new context
    up instead of stepping on the closed one.  The run equals the autograd-path run."""
    from explainn_amd import ExplaiNN, get_loss, get_metrics, get_optimizer
    from explainn_amd.selene import Trainer
    from explainn_amd.train import _get_data_loader
    rng = np.random.default_rng(3)
    L, U, k = 40, 4, 9
    codes = rng.integers(0, 4, size=(221, L))
    x = np.zeros((221, 4, L), dtype=np.float32)
    for a in range(4):
        x[:, a, :] = codes == a
    y = (rng.random((221, 1)) > 0.5).astype(np.float32)
    results = []
    for fused in (True, False):
        torch.manual_seed(1)
        model = ExplaiNN(U, k, L, 1)
        model.dropout_p = 0.0
        loaders = {"train": _get_data_loader(x[:101], y[:101], 100, shuffle=False),
                   "validation": _get_data_loader(x[101:], y[101:], 100, shuffle=False)}
        assert loaders["train"].batch_size == 99 and loaders["validation"].batch_size == 100
        crit = get_loss("binary") if fused else torch.nn.BCEWithLogitsLoss(pos_weight=torch.ones(1))
        out = tmp_path / ("fused" if fused else "autograd")
        tr = Trainer(model, loaders, crit, get_metrics("binary"),
                     get_optimizer(model.parameters(), 0.003), max_steps=6, patience=100,
                     report_stats_every_n_steps=2, output_dir=str(out), use_cuda=True,
                     logging_verbosity=0)
        assert tr._fused_step_available() == fused
        tr.train_and_validate()
        results.append((np.loadtxt(out / "train.txt", skiprows=1),
                        np.loadtxt(out / "validation.txt", skiprows=1, usecols=0)))
    assert results[0][0].shape == (3,)
    assert np.abs(results[0][0] - results[1][0]).max() < 1e-5
    assert np.abs(results[0][1] - results[1][1]).max() < 1e-4


# ---- base-code input pipeline (explainn_amd/loader.py) ---------------------------
def test_vectorised_readers_match_the_reference_encoding(tmp_path):
    """TSV / FASTA -> codes with one table lookup over the joined text: the codes expand to exactly
    the one-hot the reference's `one_hot_encode` produced (tests/golden/encoding.npz) and to what
    `_get_seqs_labels_ids` builds (train.py:266-284), reverse-complement augmentation included."""
    import gzip
    from explainn_amd import loader as ld
    from explainn_amd import sequence as sq
    from explainn_amd.train import _get_seqs_labels_ids
    z = np.load(GOLDEN + "/encoding.npz", allow_pickle=False)
    for i in range(5):
        q = str(z["seq%d" % i])
        assert np.array_equal(sq.codes_to_one_hot(ld.codes_from_strings([q]), float)[0], z["enc%d" % i])
    rng = np.random.default_rng(0)
    seqs = ["".join(rng.choice(list("ACGTNacgtRY"), size=37)) for _ in range(23)]
    tsv = tmp_path / "d.tsv"
    tsv.write_text("".join("s%d\t%s\t%d\t%.2f\n" % (i, s, i % 2, i / 7) for i, s in enumerate(seqs)))
    codes, labels, ids = ld.read_tsv_codes(str(tsv))
    ref_x, ref_y, ref_ids = _get_seqs_labels_ids(str(tsv))
    assert codes.dtype == np.uint8 and codes.shape == (23, 37) and labels.dtype == np.float32
    assert np.array_equal(sq.codes_to_one_hot(codes, float), ref_x)
    assert np.allclose(labels, ref_y) and list(ids) == list(ref_ids)
    # FASTA: multi-line records, gzip, CRLF
    fa = tmp_path / "d.fa.gz"
    with gzip.open(fa, "wt") as fh:
        for i, s in enumerate(seqs):
            fh.write(">id%d some description\r\n%s\n%s\n" % (i, s[:20], s[20:]))
    fcodes, fids = ld.read_fasta_codes(str(fa))
    assert np.array_equal(fcodes, codes) and list(fids) == ["id%d" % i for i in range(23)]
    with pytest.raises(ValueError):
        ld.codes_from_strings(["ACG", "AC"])
    # the augmented loader yields the batches the reference's loader yields, as codes
    ref_x2, ref_y2, _ = _get_seqs_labels_ids(str(tsv), reverse_complement=True)
    from torch.utils.data import DataLoader, TensorDataset
    for shuffle in (False, True):
        torch.manual_seed(9)
        ours = ld.CodesLoader(codes, labels, batch_size=12, shuffle=shuffle, reverse_complement=True)
        got = [b for _ in range(2) for b in ours]
        torch.manual_seed(9)
        ref = DataLoader(TensorDataset(torch.Tensor(ref_x2), torch.Tensor(ref_y2)), ours.batch_size,
                         shuffle=shuffle)
        want = [b for _ in range(2) for b in ref]
        assert len(ours.dataset) == 46 and len(got) == len(want) == 2 * len(ours)
        for (gc, gy), (wx, wy) in zip(got, want):
            assert gc.dtype == torch.uint8
            assert torch.equal(torch.from_numpy(sq.codes_to_one_hot(gc.numpy())), wx) and torch.equal(gy, wy)
    assert ld.CodesLoader(codes[:13], labels[:13], batch_size=12).batch_size == 11     # 13 % 12 == 1


@pytest.mark.gpu
def test_trainer_from_codes_equals_trainer_from_one_hot(tmp_path):
    """A Trainer run fed by loader.CodesLoader (pinned staging, asynchronous copies, reverse
    complement generated per batch) is bit-identical to the run fed by the reference-style
    DataLoader over the fp32 one-hot of the same augmented data set."""
    from explainn_amd import ExplaiNN, get_loss, get_metrics, get_optimizer
    from explainn_amd import sequence as sq
    from explainn_amd.loader import CodesLoader
    from explainn_amd.selene import Trainer
    from explainn_amd.train import _get_data_loader
    z, (U, k, L, T, B, N), x, y, sd = _trainer_fixture()
    codes = z["codes"]
    yy = z["y"]
    res = []
    for mode in ("onehot", "codes"):
        model = ExplaiNN(U, k, L, T)
        model.load_state_dict(sd)
        model.dropout_p = 0.0
        if mode == "onehot":
            xa = np.append(sq.codes_to_one_hot(codes[:N]), sq.codes_to_one_hot(sq.rc_codes(codes[:N])), axis=0)
            ya = np.append(yy[:N], yy[:N], axis=0)
            xv = np.append(sq.codes_to_one_hot(codes[N:]), sq.codes_to_one_hot(sq.rc_codes(codes[N:])), axis=0)
            yv = np.append(yy[N:], yy[N:], axis=0)
            loaders = {"train": _get_data_loader(xa, ya, B, shuffle=True),
                       "validation": _get_data_loader(xv, yv, B, shuffle=False)}
        else:
            loaders = {"train": CodesLoader(codes[:N], yy[:N], B, True, True, device="cuda"),
                       "validation": CodesLoader(codes[N:], yy[N:], B, False, True, device="cuda")}
        torch.manual_seed(77)
        out = tmp_path / mode
        spe = len(loaders["train"])
        tr = Trainer(model, loaders, get_loss("binary"), get_metrics("binary"),
                     get_optimizer(model.parameters(), 0.003), max_steps=spe * 3, patience=spe * 10,
                     report_stats_every_n_steps=spe, output_dir=str(out), use_cuda=True,
                     logging_verbosity=0)
        tr.train_and_validate()
        res.append((open(out / "train.txt").read(), open(out / "validation.txt").read(),
                    {key: v.detach().cpu().clone() for key, v in model.state_dict().items()}))
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1]
    for key in res[0][2]:
        assert torch.equal(res[0][2][key], res[1][2][key]), key


@pytest.mark.parametrize("gz", [False, True])
def test_filter_weights_pickle_as_the_reference_writes_it(tmp_path, gz):
    """interpret.py:154-163 pickles `{(name, 'filterN'): filter_w.T}` with HIGHEST_PROTOCOL
    (protocol 5 goes through numpy's _frombuffer for contiguous and Fortran-order arrays);
    train.py:183-195 reads it back.  The allow-listed unpickler must accept exactly that and still
    refuse anything that is not an array container."""
    import gzip
    import pickle
    from explainn_amd.train import _load_filter_weights
    rng = np.random.default_rng(0)
    filters = [rng.standard_normal((4, 19)).astype(np.float32) for _ in range(3)]
    obj = {("model.pth.tar", "filter%d" % i): w.T for i, w in enumerate(filters)}      # (k,4), F-order views
    obj[("model.pth.tar", "filter3")] = np.ascontiguousarray(filters[0].T)             # C-order too
    path = str(tmp_path / ("weights.pkl" + (".gz" if gz else "")))
    with (gzip.open(path, "wb") if gz else open(path, "wb")) as fh:
        pickle.dump(obj, fh, protocol=pickle.HIGHEST_PROTOCOL)
    ids, ws = _load_filter_weights(path)
    assert ids == ["model.pth.tar;filter%d" % i for i in range(4)]
    for w, ref in zip(ws, filters + [filters[0]]):
        assert tuple(w.shape) == (4, 19) and np.array_equal(w.numpy(), ref)
    bad = str(tmp_path / "bad.pkl")
    with open(bad, "wb") as fh:
        pickle.dump({"a": os.system}, fh, protocol=pickle.HIGHEST_PROTOCOL)
    with pytest.raises(pickle.UnpicklingError):
        _load_filter_weights(bad)


def test_debugging_cut_comes_after_the_reverse_complement_doubling():
    """train.py:272-282: `--debugging` keeps `seqs[:1000]` of the ALREADY doubled array, i.e. 1000
    items (forward sequences first), not 1000 rows doubled to 2000."""
    from explainn_amd.loader import CodesLoader
    rng = np.random.default_rng(3)
    for n_rows in (1200, 600):
        codes = rng.integers(0, 4, (n_rows, 30)).astype(np.uint8)
        labels = np.arange(n_rows, dtype=np.float32)[:, None]
        # what the reference builds: forward rows, then their reverse complements, then the cut
        doubled = np.concatenate([codes, (3 - codes)[:, ::-1]])[:1000]
        doubled_y = np.concatenate([labels, labels])[:1000]
        loader = CodesLoader(codes[:1000], labels[:1000], batch_size=100, shuffle=False,
                             reverse_complement=True, limit=1000)
        assert len(loader.dataset) == 1000 and len(loader) == 10
        got_c = np.concatenate([c.numpy() for c, _ in loader])
        got_y = np.concatenate([y.numpy() for _, y in loader])
        assert np.array_equal(got_c, doubled) and np.array_equal(got_y, doubled_y)
