#!/usr/bin/env python3
"""Per-kernel counters of one training step from separate rocprofv3 --pmc passes (FETCH_SIZE and
WRITE_SIZE do not fit one pass, the SQ set needs two -- MI355X_MICROARCH.md, counter table).

    python tools/traffic.py <dir with fetch/ write/ sq1/ sq2/> <out.json> [round-tag] [workload]

HBM-side bytes: FETCH_SIZE / WRITE_SIZE are KiB per dispatch; gfx950 correction (same guide, HBM
section): FETCH_SIZE tallies 128-B requests at 64 B -> x2; WRITE_SIZE is exact for wide streaming
stores.  Per step = sum over this library's kernels / number of steps (= launches of the pack
kernel).  The output carries the hash of the kernel sources it was taken on; bench.py uses it only
when that hash still matches."""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = {"pack_tables_kernel": "pack_tables", "moments_kernel": "moments", "prep1_stats_kernel": "prep1_stats",
         "conv_pool_mm_kernel": "conv_pool", "qmom_kernel": "qmom", "qmom_big_kernel": "qmom",
         "prep2_kernel": "prep2", "fc_fwd_kernel": "fc_fwd", "fc_fwd_bf_kernel": "fc_fwd",
         "head_fwd_train_kernel": "head_fwd", "logits_kernel": "head_fwd", "logits_bn_kernel": "head_fwd", "loss_kernel": "loss", "head_bwd_kernel": "head_bwd",
         "passA_kernel": "passA", "mid_fused_kernel": "mid", "mid_big_kernel": "mid",
         "passB_kernel": "passB", "conv_bwd_kernel": "conv_bwd", "conv_bwd_mm_kernel": "conv_bwd",
         "fin_bwd_kernel": "fin_bwd",
         "gemm32_kernel": "head_gemm"}


def csrc_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "explainn_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as fh:
                h.update(name.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


def per_kernel(path):
    """{counter: {kernel base name: [sum, calls]}} of one pass."""
    files = glob.glob(path + "/*/*counter_collection.csv")
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    if not files:
        return out
    for r in csv.DictReader(open(files[0])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if name.startswith(("__amd", "at::")):
            continue
        e = out[r["Counter_Name"]][name]
        e[0] += float(r["Counter_Value"]); e[1] += 1
    return out


def main():
    base, out = sys.argv[1:3]
    tag = sys.argv[3] if len(sys.argv) > 3 else "r02"
    workload = sys.argv[4] if len(sys.argv) > 4 else "C2"
    passes = {p: per_kernel(os.path.join(base, p)) for p in ("fetch", "write", "sq1", "sq2")}
    ft, wt = passes["fetch"]["FETCH_SIZE"], passes["write"]["WRITE_SIZE"]
    steps = max(v[1] for k, v in ft.items() if k.startswith(("pack_onehot", "pack_tables")))
    per = collections.defaultdict(dict)
    for k, (v, _) in ft.items():
        per[STAGE.get(k, k)]["fetch_raw_bytes"] = per[STAGE.get(k, k)].get("fetch_raw_bytes", 0) + v * 1024 / steps
    for k, (v, _) in wt.items():
        per[STAGE.get(k, k)]["write_bytes"] = per[STAGE.get(k, k)].get("write_bytes", 0) + v * 1024 / steps
    for p in ("sq1", "sq2"):
        for counter, kern in passes[p].items():
            for k, (v, calls) in kern.items():
                st = per[STAGE.get(k, k)]
                st[counter] = st.get(counter, 0) + v / steps        # per step (summed over the stage's launches)
    for st in per.values():
        st["hbm_bytes"] = int(2 * st.get("fetch_raw_bytes", 0) + st.get("write_bytes", 0))
    fetch_raw = sum(st.get("fetch_raw_bytes", 0) for st in per.values())
    write = sum(st.get("write_bytes", 0) for st in per.values())
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        head = None
    res = {
        "source": "rocprofv3 --kernel-trace --pmc <one counter set per pass> over `python3 bench.py --workload %s "
                  "--no-cpu-baseline --skip-optimizer --skip-stage-times --steps 5 --warmup 2`, all launches of "
                  "this library, per training step; raw CSVs: profiles/%s_pmc_*.csv" % (workload, tag),
        "workload": workload, "csrc_sha": csrc_sha(), "head": head, "steps_profiled": steps,
        "kernel_stages": sorted(per),
        "fetch_raw_MB_per_step": round(fetch_raw / 1e6, 1), "write_MB_per_step": round(write / 1e6, 1),
        "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 64 B per 128-B request, MI355X_MICROARCH.md)",
        "traffic_bytes_per_step": int(round(2 * fetch_raw + write, -5)),
        "units": "SQ_* counters: summed over the stage's launches of one step; SQ_WAVE_CYCLES / SQ_WAIT_* / "
                 "SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles, SQ_INSTS_* wave-instructions",
        "per_kernel": {k: {c: (round(v, 1) if isinstance(v, float) else v) for c, v in sorted(st.items())}
                       for k, st in sorted(per.items(), key=lambda kv: -kv[1].get("hbm_bytes", 0))},
    }
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("fetch_raw_MB_per_step", "write_MB_per_step",
                                          "traffic_bytes_per_step", "csrc_sha")}))


if __name__ == "__main__":
    main()
