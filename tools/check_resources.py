#!/usr/bin/env python3
"""Build-time register / scratch check ("spills are HBM traffic").

The Makefile compiles every .hip with -Rpass-analysis=kernel-resource-usage and keeps the
remarks in build/<file>.res.  This script parses them and

  * FAILS (exit 1) when a kernel of the headline training step (config C2: pooled length bucket
    26, kernel size 19, and every non-templated stage kernel) has ScratchSize > 0 or any VGPR/SGPR
    spill -- a spilled hot kernel once cost 92 MB of scratch traffic per launch (r01);
  * prints every other kernel that spills (large-n / rare instantiations) as a warning;
  * with --table prints name, VGPRs, SGPRs, LDS, occupancy of the C2 step's kernels.

usage: check_resources.py build_dir [--table]
"""
import glob
import os
import re
import subprocess
import sys

FIELDS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch",
          "Occupancy [waves/SIMD]": "occ", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
          "LDS Size [bytes/block]": "lds"}

# kernels one C2 training step launches (U = 300, k = 19, L = 200 -> n = 26, T = 1, B = 1024):
# demangled-name patterns.  Templated kernels are matched on their C2 instantiation.
C2_STEP = [
    r"^pack_tables_kernel", r"^moments_kernel", r"^prep1_stats_kernel<true>", r"^conv_pool_mm_kernel<5, 2,",
    r"^qmom_kernel<26>", r"^prep2_kernel<true>", r"^fc_fwd_bf_kernel<26, 2>", r"^logits_bn_kernel",
    r"^head_bwd_kernel<true, false>", r"^passA_kernel<26, false>", r"^mid_fused_kernel", r"^passB_kernel<26>",
    r"^conv_bwd_mm_kernel<19>", r"^fin_bwd_kernel", r"adam_kernel",
]


# Register cliffs measured in the pipeline: 1024-thread blocks of which TWO must
# share a CU (300 units on 256 CUs take two rounds otherwise) -- 64 registers, not one more.
VGPR_CAP = {r"^mid_fused_kernel": 64, r"^prep2_kernel<true>": 64}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return [re.sub(r"^void ", "", l.replace("(anonymous namespace)::", "").split("(")[0]) for l in out.splitlines()]


def parse(path):
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = {"mangled": m.group(1), "file": os.path.basename(path)[:-4]}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in FIELDS:
            cur[FIELDS[m.group(1).strip()]] = int(m.group(2))
    return rows


def main():
    build = sys.argv[1]
    rows = []
    for f in sorted(glob.glob(os.path.join(build, "*.res"))):
        rows += parse(f)
    if not rows:
        print("check_resources: no *.res files under %s (build with the Makefile)" % build)
        return 1
    for r, name in zip(rows, demangle([r["mangled"] for r in rows])):
        r["name"] = name
    # what costs memory traffic: scratch and VGPR spills.  SGPR spills go to VGPR lanes
    # (v_writelane / v_readlane), no memory involved: reported in the table, not an error
    hard = lambda r: r.get("scratch", 0) > 0 or r.get("vgpr_spill", 0) > 0
    spilled = lambda r: hard(r) or r.get("sgpr_spill", 0) > 0
    hot = [r for r in rows if any(re.search(p, r["name"]) for p in C2_STEP)]
    missing = [p for p in C2_STEP if not any(re.search(p, r["name"]) for r in rows)]
    bad = [r for r in hot if hard(r)]
    others = [r for r in rows if spilled(r) and r not in hot]
    if "--table" in sys.argv:
        print("%-44s %5s %5s %7s %4s  %s" % ("kernel (C2 step)", "VGPR", "SGPR", "LDS", "occ", "scratch/spills"))
        for r in hot:
            print("%-44s %5d %5d %7d %4d  %d B/lane, %d sgpr, %d vgpr" % (
                r["name"][:44], r.get("vgpr", 0) + r.get("agpr", 0), r.get("sgpr", 0), r.get("lds", 0),
                r.get("occ", 0), r.get("scratch", 0), r.get("sgpr_spill", 0), r.get("vgpr_spill", 0)))
    # SGPR spills alone go to VGPR lanes (v_writelane), not to memory: counted, not listed
    for r in others:
        if r.get("scratch", 0) > 0:
            print("check_resources: note: %s (%s) uses scratch: %d B/lane, %d sgpr, %d vgpr spilled" % (
                r["name"], r["file"], r.get("scratch", 0), r.get("sgpr_spill", 0), r.get("vgpr_spill", 0)))
    rc = 0
    for r in rows:
        for pat, cap in VGPR_CAP.items():
            if re.search(pat, r["name"]) and r.get("vgpr", 0) + r.get("agpr", 0) > cap:
                print("check_resources: ERROR: %s uses %d registers, over its cap of %d (two blocks per CU)"
                      % (r["name"], r.get("vgpr", 0) + r.get("agpr", 0), cap))
                rc = 1
    for p in missing:
        print("check_resources: ERROR: no kernel matches the C2-step pattern %r (renamed? update C2_STEP)" % p)
        rc = 1
    for r in bad:
        print("check_resources: ERROR: hot kernel %s (%s) spills: scratch %d B/lane, %d sgpr, %d vgpr" % (
            r["name"], r["file"], r.get("scratch", 0), r.get("sgpr_spill", 0), r.get("vgpr_spill", 0)))
        rc = 1
    print("check_resources: %d kernels, %d in the C2 step, %d of those spill; elsewhere %d use scratch, "
          "%d more spill SGPRs only" % (len(rows), len(hot), len(bad),
                                        sum(1 for r in others if r.get("scratch", 0) > 0),
                                        sum(1 for r in others if r.get("scratch", 0) == 0)))
    return rc


if __name__ == "__main__":
    sys.exit(main())
