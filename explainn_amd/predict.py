"""Prediction entry points (reference: explainn/predict.py).

`_load_model` (predict.py:133-151) rebuilds the model from a checkpoint's `options` and
`state_dict`; `predict` is the loop of predict.py:75-94: eval-mode logits of every sequence and of
its reverse complement -> [Fwd, Rev, Mean, Max], shape (N, n_features, 4) float64.
"""
import argparse
import sys

import numpy as np
import torch

from .architectures import ExplaiNN
from .selene import _load_checkpoint_file
from .sequence import one_hot_encode_many


def _load_model(model_file):
    device = "cuda" if torch.cuda.is_available() else "cpu"
    selene_dict = _load_checkpoint_file(model_file)
    o = selene_dict["options"]
    model = ExplaiNN(o["cnn_units"], o["kernel_size"], o["sequence_length"], o["n_features"],
                     o["weights_file"])
    model.load_state_dict(selene_dict["state_dict"])
    model.to(device)
    model.eval()
    return model


_CHUNK = 4096        # sequences per device pass of predict()


def predict(model, Xs, batch_size=100, apply_sigmoid=False):
    """predict.py:75-94: eval-mode logits of both strands and their mean and max, (N, T, 4) float64
    in the order [Fwd, Rev, Mean, Max].

    Xs: (N,4,L) one-hot (numpy or tensor, host) -- or (N,L) uint8 base codes
    (sequence.encode_codes_many), in which case the reverse strand is read on the fly from the same
    bytes.  In eval mode a sequence's output does not depend on what else is in its batch, so the
    device passes use max(batch_size, 4096) sequences whatever `batch_size` says (the reference's
    default of 100 would leave the GPU waiting on the host), with one transfer back per pass; the
    numbers are the same as with batches of `batch_size`."""
    from .architectures import BaseCodes
    device = model.final.weight.device
    Xs_np = Xs if torch.is_tensor(Xs) else np.asarray(Xs)
    as_codes = (Xs_np.dtype in (np.uint8, torch.uint8)) and Xs_np.ndim == 2
    if as_codes:
        data = torch.as_tensor(np.ascontiguousarray(Xs_np)) if not torch.is_tensor(Xs_np) else Xs_np
    else:
        data = torch.as_tensor(Xs_np, dtype=torch.float32)
    chunk = max(int(batch_size), _CHUNK)
    out = np.empty((len(data), model._options["n_features"], 4))
    # the folded eval tables are built once for the whole loop (nothing writes parameters here);
    # input validation is settled by ONE read of the sticky device flag after the last pass.
    # The two strands of a chunk are independent batches: the reverse strand runs on a replica of the
    # model (same tensors, own device context: ExplaiNN.eval_replica) on a second stream, so that the
    # launches of one strand fill the gaps between the dependent launches of the other.
    rep = model.eval_replica()
    cur = torch.cuda.current_stream(device)
    if model._rt.side_stream is None:
        model._rt.side_stream = torch.cuda.Stream(device)
    side = model._rt.side_stream
    with torch.no_grad(), model.eval_cache(), rep.eval_cache():
        for i in range(0, len(data), chunk):
            xb = data[i:i + chunk].to(device)
            if as_codes:
                xf, xr = BaseCodes(xb), BaseCodes(xb, reverse_complement=True)
            else:
                xf, xr = xb, torch.flip(xb, dims=(1, 2))       # rc_one_hot_encoding: both axes
            side.wait_stream(cur)                               # the chunk is on the device
            with torch.cuda.stream(side):
                rev = rep(xr)
            fwd = model(xf)
            cur.wait_stream(side)
            rev.record_stream(cur)
            # numpy's float32 mean of two values is (a + b) / 2 in float32, as here
            both = torch.stack((fwd, rev, (fwd + rev) / 2, torch.maximum(fwd, rev)), dim=2)
            out[i:i + both.shape[0]] = both.cpu().numpy()
    if model.validate_input:
        rep.check_input()
    if model.validate_input:
        model.check_input()
    if apply_sigmoid:
        out = torch.sigmoid(torch.Tensor(out)).numpy()
    return out


def _read_fasta(path):
    ids, seqs, cur = [], [], []
    opener = open
    if path.endswith(".gz"):
        import gzip
        opener = gzip.open
    with opener(path, "rt") as fh:
        for line in fh:
            line = line.strip()
            if line.startswith(">"):
                if cur:
                    seqs.append("".join(cur)); cur = []
                ids.append(line[1:].split()[0])
            elif line:
                cur.append(line)
    if cur:
        seqs.append("".join(cur))
    return np.array(ids), seqs


def main(argv=None):
    """predict.py:50-118: FASTA -> long-format TSV (SeqId, Class, Fwd, Rev, Mean, Max)."""
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("model_file"); ap.add_argument("fasta_file")
    ap.add_argument("-b", "--batch-size", type=int, default=2 ** 6)
    ap.add_argument("-o", "--output-file")
    ap.add_argument("-s", "--apply-sigmoid", action="store_true")
    args = ap.parse_args(argv)
    import pandas as pd
    ids, seqs = _read_fasta(args.fasta_file)
    model = _load_model(args.model_file)
    preds = predict(model, one_hot_encode_many(seqs), args.batch_size, args.apply_sigmoid)
    dfs = []
    for i in range(model._options["n_features"]):
        df = pd.DataFrame(preds[:, i, :], columns=["Fwd", "Rev", "Mean", "Max"])
        df["SeqId"] = ids
        df["Class"] = i
        dfs.append(df)
    df = pd.concat(dfs)[["SeqId", "Class", "Fwd", "Rev", "Mean", "Max"]].reset_index(drop=True)
    df.to_csv(args.output_file if args.output_file else sys.stdout, sep="\t", index=False)


if __name__ == "__main__":
    main()
