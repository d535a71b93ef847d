"""CPU restatement of the reference's filter -> PWM path (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and tools/make_golden.py import this module; the product
(explainn_amd/interpret.py + csrc/interpret.hip) never does.

What it restates (all numpy, small inputs):
  * test.py:128-166   `_get_acts_outs_preds`   -- float16 storage of activations/outputs/predictions
  * interpret.py:310-361 `_get_well_predicted_sequences`
  * interpret.py:363-373 `_get_act_thresholds`  -- 0.5 * max activation over well-predicted sequences
  * interpret.py:375-429 `_get_sites`           -- start positions with activation > threshold, in
                                                   (strand, sequence, position) order, capped at 1e6
  * interpret.py:431-459 `_sites_to_motif`      -- A/C/G/T counts per site column (Bio.motifs counts;
                                                   biopython is absent here, the count is restated)
  * interpret.py:485-490 `_filter_filter_importances` and the loop at interpret.py:176-183

Pinning: the activations/outputs/predictions that feed it in tests/golden/pfm_*.npz come from the
imported reference model (tools/make_golden.py); the post-processing above has no reference-held
fixture (the reference has no tests) -- it is pinned by reading, not by vectors ("parity unpinned"
for the site/PFM bookkeeping itself, as DESIGN.md records).
"""
import numpy as np

from . import explainn_oracle as eo

SITE_CAP = 1000000            # interpret.py:423-424


def acts_outs_preds(sd, x):
    """test.py:128-166: (N,U,Lo) activations, (N,U) unit outputs, (N,T) predictions, float16."""
    logits, cache, _ = eo.forward(sd, x, training=False, dtype=np.float32, return_cache=True)
    return (cache["acts"].astype(np.float16), cache["o"].astype(np.float16),
            logits.astype(np.float16))


def _halves(arr, strand):
    """test.py:198-203."""
    h = len(arr) // 2
    return arr[:h] if strand in ("fwd", "+") else arr[h:]


def _sigmoid16(a16):
    """torch.sigmoid on a float16 CPU tensor: computed in fp32, rounded to fp16."""
    a = a16.astype(np.float32)
    return (1.0 / (1.0 + np.exp(-a))).astype(np.float16)


def well_predicted_sequences(preds, labels, input_data, rev_complement=False):
    """interpret.py:310-361.  preds float16 (N,T), labels (N,T); returns sorted unique indices
    (into the forward half when rev_complement)."""
    n = .05
    if rev_complement:
        fwd, rev = _halves(preds, "fwd"), _halves(preds, "rev")
        p = np.empty(fwd.shape)
        ys = _halves(labels, "fwd")
        for i in range(p.shape[1]):
            p[:, i] = np.mean([fwd[:, i], rev[:, i]], axis=0)
            if input_data == "binary":
                # sigmoid of a float64 column (p is float64 here, interpret.py:322-328)
                p[:, i] = 1.0 / (1.0 + np.exp(-p[:, i]))
    else:
        p = _sigmoid16(preds) if input_data == "binary" else preds
        ys = labels
    if input_data == "binary":
        ok = (ys == (p > .5).astype(int))
        idxs = np.where(ok.all(axis=1))[0]
    else:
        m = int(max(ys.shape) * n)
        idxs_ys = np.argsort(-ys.flatten(), kind="stable")[:m]
        idxs_p = np.argsort(-p.flatten(), kind="stable")[:m]
        idxs = np.intersect1d(idxs_ys, idxs_p)
    return np.asarray(idxs, dtype=np.int64)


def act_thresholds(acts, idxs, rev_complement=False):
    """interpret.py:363-373 (float16 in, float16 out)."""
    if rev_complement:
        sel = np.concatenate((_halves(acts, "fwd")[idxs], _halves(acts, "rev")[idxs]))
    else:
        sel = acts[idxs]
    return 0.5 * np.amax(sel, axis=(0, 2))


def site_pfms(codes, acts, idxs, thresholds, k, rev_complement=False, cap=SITE_CAP):
    """interpret.py:375-459 without the FASTA round trip: for every unit, the (k,4) A/C/G/T counts
    of the sites `_get_sites` would write (an N inside a site counts for no letter) and the number of
    sites.  codes: uint8 (N,L), 0..3 = ACGT, 4 = N."""
    N, U, Lo = acts.shape
    pfm = np.zeros((U, k, 4), dtype=np.int64)
    nsites = np.zeros(U, dtype=np.int64)
    for u in range(U):
        count = 0
        done = False
        for strand in ("+", "-"):
            if rev_complement:
                c_arr, a_arr = _halves(codes, strand), _halves(acts, strand)
            else:
                c_arr, a_arr = codes, acts
            for i in idxs:
                starts = np.where(a_arr[i, u, :] > thresholds[u])[0]
                for j in starts:
                    site = c_arr[i, j:j + k]
                    for t in range(k):
                        if site[t] < 4:
                            pfm[u, t, site[t]] += 1
                    count += 1
                    if count == cap:
                        done = True
                        break
                if done:
                    break
            if done or not rev_complement:
                break
        nsites[u] = count
    return pfm, nsites


def filter_importances(outs, final_w, idxs, acts, thresholds):
    """interpret.py:176-183 + 485-490: per unit, the (T, n_selected) importances outs*weight of the
    well-predicted sequences that have at least one position above the unit's threshold.
    Returns a list of (selected indices, importances) per unit."""
    res = []
    imps = np.array([np.multiply(outs, final_w[t, :]) for t in range(final_w.shape[0])])
    for u in range(outs.shape[1]):
        rows = np.where(acts[:, u, :] > thresholds[u])[0]
        sel = np.intersect1d(idxs, rows)
        res.append((sel, imps[:, sel, u]))
    return res
