"""Filter -> PWM interpretation on the device.

Mirrors the part of the reference's interpret.py that turns a trained ExplaiNN into one position
frequency matrix per filter (interpret.py:160-235):

    acts, outs, preds = _get_acts_outs_preds(model, loader)         test.py:128-166
    idxs       = _get_well_predicted_sequences(preds, labels, ...)  interpret.py:310-361
    thresholds = _get_act_thresholds(acts, idxs, rc)                interpret.py:363-373
    sites      = _get_sites(...); motif = _sites_to_motif(sites)    interpret.py:375-459
    imps       = _filter_filter_importances(...)                    interpret.py:485-490

The reference materialises `acts` as a dense float16 (N,U,Lo) host array (11 GB for 100 K sequences
at 300 units / 200 bp) and walks it with Python loops that write FASTA files.  Here the activations
are recomputed inside two HIP passes over the packed base codes (csrc/interpret.hip) and only the
(U,) maxima, the (U,k,4) count matrices and a (N,U) "has a site" bit ever leave the kernel; results
are the same numbers (float16 rounding of the activations included).

`_get_acts_outs_preds` and `_get_well_predicted_sequences` keep the reference's names and argument
meaning; `filter_pwms` replaces the thresholds -> sites -> motif chain.
"""
import argparse
import os
import time

import numpy as np
import torch

SITE_CAP = 1000000          # interpret.py:423-424: a filter with 1e6 sites is "way too ubiquitous"


def _get_fwd_rev(arr, strand):
    """test.py:198-203."""
    half = len(arr) // 2
    if strand in ("fwd", "+"):
        return arr[:half]
    if strand in ("rev", "-"):
        return arr[half:]
    raise ValueError("strand must be fwd/+ or rev/-")


def _batches(Xs, batch_size):
    for i in range(0, len(Xs), batch_size):
        yield i, Xs[i:i + batch_size]


def _as_tensor(Xs):
    return Xs if torch.is_tensor(Xs) else torch.as_tensor(np.asarray(Xs), dtype=torch.float32)


def _get_outs_preds(exp_model, Xs, batch_size=100):
    """The (N,U) unit outputs and (N,T) predictions of test.py:128-166, float16 like there.
    Eval-mode outputs do not depend on batch composition, so the device passes take at least 4096
    sequences whatever batch_size says."""
    batch_size = max(int(batch_size), 4096)
    dev = exp_model.final.weight.device
    U, T = exp_model._options["cnn_units"], exp_model._options["n_features"]
    outputs = np.zeros((len(Xs), U), dtype=np.float16)
    predictions = np.zeros((len(Xs), T), dtype=np.float16)
    Xs = _as_tensor(Xs)
    with torch.no_grad(), exp_model.eval_cache():
        for i, xb in _batches(Xs, batch_size):
            outs = exp_model.linears(xb.to(dev))
            outputs[i:i + len(xb)] = outs.cpu().numpy()
            predictions[i:i + len(xb)] = exp_model.final(outs).cpu().numpy()
    return outputs, predictions


def _get_acts_outs_preds(exp_model, data_loader):
    """test.py:128-166, kept for callers that want the dense float16 activation array."""
    o = exp_model._options
    N = len(data_loader.dataset)
    Lo = o["sequence_length"] - o["kernel_size"] + 1
    activations = np.zeros((N, o["cnn_units"], Lo), dtype=np.float16)
    outputs = np.zeros((N, o["cnn_units"]), dtype=np.float16)
    predictions = np.zeros((N, o["n_features"]), dtype=np.float16)
    dev = exp_model.final.weight.device
    idx = 0
    with torch.no_grad(), exp_model.eval_cache():
        for Xs, _ in data_loader:
            Xs = Xs.to(dev)
            outs = exp_model.linears(Xs)
            outputs[idx:idx + len(Xs)] = outs.cpu().numpy()
            predictions[idx:idx + len(Xs)] = exp_model.final(outs).cpu().numpy()
            activations[idx:idx + len(Xs)] = exp_model.linears[:3](Xs).cpu().numpy()
            idx += len(Xs)
    return activations, outputs, predictions


def _get_well_predicted_sequences(preds, labels, input_data, rev_complement=False):
    """interpret.py:310-361 (host logic on (N,T) arrays).  Binary data: sequences whose thresholded
    prediction equals the label for every task; otherwise the intersection of the top-5 % labels and
    top-5 % predictions."""
    frac = .05
    if rev_complement:
        fwd, rev = _get_fwd_rev(preds, "fwd"), _get_fwd_rev(preds, "rev")
        ys = _get_fwd_rev(labels, "fwd")
        p = np.empty(fwd.shape)
        for t in range(p.shape[1]):
            p[:, t] = np.mean([fwd[:, t], rev[:, t]], axis=0)
            if input_data == "binary":
                p[:, t] = torch.sigmoid(torch.from_numpy(p[:, t])).numpy()
    else:
        p = torch.sigmoid(torch.from_numpy(preds)).numpy() if input_data == "binary" else preds
        ys = labels
    if input_data == "binary":
        agree = ys == (p > .5).astype(int)
        return np.where(agree.all(axis=1))[0]
    top = int(max(ys.shape) * frac)
    return np.intersect1d(np.argsort(-ys.flatten())[:top], np.argsort(-p.flatten())[:top])


def filter_pwms(exp_model, Xs, idxs, rev_complement=False, batch_size=1024, site_cap=SITE_CAP):
    """thresholds -> sites -> count matrices for every filter, on the device.

    Xs: (N,4,L) one-hot, forward strands followed (when rev_complement) by their reverse complements
    in the same order, as train._get_seqs_labels_ids lays them out; idxs: the well-predicted
    sequence indices (into the forward half when rev_complement).

    Returns dict(thresholds float16 [U], pfm int64 (U,k,4) rows A,C,G,T per site column,
    nsites int64 [U], hit bool (N,U))."""
    o = exp_model._options
    U, k = o["cnn_units"], o["kernel_size"]
    dev = exp_model.final.weight.device
    Xs = _as_tensor(Xs)
    N = len(Xs)
    half = N // 2 if rev_complement else N
    sel = np.zeros(N, dtype=np.uint8)
    idxs = np.asarray(idxs, dtype=np.int64)
    sel[idxs] = 1
    if rev_complement:
        sel[idxs + half] = 1
    select = torch.from_numpy(sel).to(dev)
    was_training = exp_model.training
    exp_model.eval()
    try:
        with exp_model.eval_cache():
            unit_max = torch.zeros(U, device=dev, dtype=torch.float32)
            for i, xb in _batches(Xs, batch_size):
                exp_model.filter_act_max(xb.to(dev), unit_max, select[i:i + len(xb)])
            # interpret.py:373: 0.5 * amax of a float16 array stays float16
            thresholds = (0.5 * unit_max.cpu().numpy().astype(np.float16)).astype(np.float16)
            thr_dev = torch.from_numpy(thresholds.astype(np.float32)).to(dev)
            site_total = torch.zeros(U, device=dev, dtype=torch.int32)
            pfm = torch.zeros(U, k, 4, device=dev, dtype=torch.int32)
            hit = np.zeros((N, U), dtype=bool)
            # forward strand first, then the reverse strand (interpret.py:385-429); a batch never
            # straddles the two halves, so site ranks follow the reference's order
            bounds = [(0, half)] + ([(half, N)] if rev_complement else [])
            for lo, hi in bounds:
                for i in range(lo, hi, batch_size):
                    j = min(i + batch_size, hi)
                    h = exp_model.filter_sites(Xs[i:j].to(dev), thr_dev, site_total, pfm, select[i:j],
                                               site_cap=site_cap, want_hit=True)
                    hit[i:j] = h.cpu().numpy().astype(bool)
        if exp_model.validate_input:
            exp_model.check_input()
    finally:
        exp_model.train(was_training)
    return {"thresholds": thresholds, "pfm": pfm.cpu().numpy().astype(np.int64),
            "nsites": site_total.cpu().numpy().astype(np.int64), "hit": hit}


def filter_importances(outs, final_weight, idxs, hit):
    """interpret.py:176-183 + 485-490: for each unit the (T, n) importances outs*weight of the
    well-predicted sequences with at least one position above the unit's threshold.  `hit` is
    filter_pwms' (N,U) matrix (it replaces `np.where(acts > threshold)[0]` on the dense array)."""
    res = []
    for u in range(outs.shape[1]):
        sel = np.intersect1d(idxs, np.where(hit[:, u])[0])
        imps = np.array([np.multiply(outs[sel, u], final_weight[t, u])
                         for t in range(final_weight.shape[0])])
        res.append((sel, imps))
    return res


def format_jaspar(pfm_u, matrix_id, name):
    """One motif in the JASPAR text layout Bio.motifs writes for `format(motif, "jaspar")`
    (interpret.py:228-231).  biopython is not available in this image, so the layout follows its
    published format ('>id name' then 'A [ %6.2f ...]' rows) and is not pinned by a fixture."""
    lines = [">%s %s\n" % (matrix_id, name)]
    for a, letter in enumerate("ACGT"):
        lines.append("%s [%s]\n" % (letter, " ".join("%6.2f" % v for v in pfm_u[:, a])))
    return "".join(lines)


def interpret(exp_model, seqs, labels, name, output_dir="./", batch_size=100, rev_complement=False,
              input_data=None):
    """The filter-level part of interpret.py's main (interpret.py:128-235): output-layer weights,
    filter importances and one JASPAR motif per filter, written under output_dir."""
    import pandas as pd
    if input_data is None:
        input_data = "binary" if np.unique(labels[:, 0]).size == 2 else "linear"
    os.makedirs(os.path.join(output_dir, "motifs"), exist_ok=True)
    weights = exp_model.final.weight.detach().cpu().numpy()
    U = weights.shape[1]
    rows = [["filter%d" % u] + w.tolist() for u, w in enumerate(weights.T)]
    pd.DataFrame(rows, columns=["filter"] + list(range(weights.shape[0]))).to_csv(
        os.path.join(output_dir, "output-layer-weights.tsv"), sep="\t", index=False)
    outs, preds = _get_outs_preds(exp_model, seqs, batch_size)
    idxs = _get_well_predicted_sequences(preds, labels, input_data, rev_complement)
    res = filter_pwms(exp_model, seqs, idxs, rev_complement, batch_size=max(batch_size, 4096))
    data = []
    for u, (_, imps) in enumerate(filter_importances(outs, weights, idxs, res["hit"])):
        data.extend([["filter%d" % u] + col.tolist() for col in imps.T])
    cols = ["filter"] + list(range(weights.shape[0]))
    df = pd.DataFrame(data, columns=cols)
    tsv = os.path.join(output_dir, "filter-importances.tsv")
    df.to_csv(tsv + ".gz", sep="\t", index=False, compression="gzip")
    df = df.groupby(["filter"]).median().sort_values([cols[-1]], ascending=False)
    df.reset_index(inplace=True)
    df.to_csv(tsv, sep="\t", index=False)
    for u in range(U):
        with open(os.path.join(output_dir, "motifs", "filter%d.jaspar" % u), "wt") as fh:
            if res["nsites"][u] > 0:          # the reference leaves the file empty when no site
                fh.write(format_jaspar(res["pfm"][u], "filter%d" % u, name))
    return res


def main(argv=None):
    from .predict import _load_model
    from .train import _get_seqs_labels_ids
    ap = argparse.ArgumentParser(description="Filter PWMs and importances of a trained ExplaiNN "
                                             "(device-side counterpart of the reference's interpret.py)")
    ap.add_argument("model_file")
    ap.add_argument("training_file")
    ap.add_argument("-b", "--batch-size", type=int, default=100)
    ap.add_argument("-d", "--debugging", action="store_true")
    ap.add_argument("-n", "--name", required=True)
    ap.add_argument("-o", "--output-dir", default="./")
    ap.add_argument("-r", "--rev-complement", action="store_true")
    ap.add_argument("-t", "--time-me", action="store_true")
    args = ap.parse_args(argv)
    t0 = time.time()
    seqs, labels, _ = _get_seqs_labels_ids(args.training_file, args.debugging, args.rev_complement)
    model = _load_model(args.model_file)
    interpret(model, seqs, labels, args.name, args.output_dir, args.batch_size, args.rev_complement)
    if args.time_me:
        with open(os.path.join(args.output_dir, "time-interpret.py.txt"), "wt") as fh:
            fh.write("%.2f seconds" % (time.time() - t0))


if __name__ == "__main__":
    main()
