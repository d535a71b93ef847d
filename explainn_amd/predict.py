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


def predict(model, Xs, batch_size=100, apply_sigmoid=False):
    """Xs: (N,4,L) one-hot (numpy or tensor, host) -- or (N,L) uint8 base codes
    (sequence.encode_codes_many), in which case the reverse strand is read on the fly from the same
    bytes.  Returns (N, T, 4) float64."""
    device = model.final.weight.device
    Xs_np = np.asarray(Xs)
    if Xs_np.dtype == np.uint8 and Xs_np.ndim == 2:
        from .architectures import BaseCodes
        codes = torch.from_numpy(np.ascontiguousarray(Xs_np))
        out = np.empty((len(codes), model._options["n_features"], 4))
        with torch.no_grad():
            for i in range(0, len(codes), batch_size):
                cb = codes[i:i + batch_size].to(device)
                fwd = model(BaseCodes(cb)).cpu().numpy()[:, :, None]
                rev = model(BaseCodes(cb, reverse_complement=True)).cpu().numpy()[:, :, None]
                fr = np.concatenate((fwd, rev), axis=2)
                out[i:i + fwd.shape[0]] = np.concatenate(
                    (fwd, rev, fr.mean(axis=2, keepdims=True), fr.max(axis=2, keepdims=True)), axis=2)
        return torch.sigmoid(torch.Tensor(out)).numpy() if apply_sigmoid else out
    Xs = torch.as_tensor(Xs_np, dtype=torch.float32)
    out = np.empty((len(Xs), model._options["n_features"], 4))
    with torch.no_grad():
        for i in range(0, len(Xs), batch_size):
            fwd_x = Xs[i:i + batch_size].to(device)
            rev_x = torch.flip(fwd_x, dims=(1, 2))            # rc_one_hot_encoding: both axes
            fwd = model(fwd_x).cpu().numpy()[:, :, None]
            rev = model(rev_x).cpu().numpy()[:, :, None]
            fr = np.concatenate((fwd, rev), axis=2)
            out[i:i + fwd.shape[0]] = np.concatenate(
                (fwd, rev, fr.mean(axis=2, keepdims=True), fr.max(axis=2, keepdims=True)), axis=2)
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
