#!/usr/bin/env python3
"""Copy one tools/final_profile.sh run (gpurun_out/<run>/) into profiles/<tag>_* and append the
trace-vs-HIP-event reconciliation to the kernel summary.   python tools/install_profiles.py r01_k r01_final"""
import csv, glob, json, shutil, sys
run, tag = sys.argv[1], sys.argv[2]
S = "gpurun_out/" + run
def cp(pattern, dst):
    shutil.copy(glob.glob(S + "/" + pattern)[0], "profiles/%s_%s" % (tag, dst))
cp("bench.json", "bench.json"); cp("stats/*/*kernel_stats.csv", "kernel_stats.csv")
cp("kernel_summary.txt", "kernel_summary.txt"); cp("fetch/*/*counter_collection.csv", "pmc_FETCH_SIZE.csv")
cp("write/*/*counter_collection.csv", "pmc_WRITE_SIZE.csv"); cp("sq/*/*counter_collection.csv", "pmc_SQ.csv")
cp("pmc_sq_summary.txt", "pmc_SQ_summary.txt")
t = open(S + "/traffic.json").read().replace(run + "_pmc", tag + "_pmc")
open("profiles/%s_traffic.json" % tag, "w").write(t)
d = json.loads(open("profiles/%s_bench.json" % tag).read().strip().splitlines()[-1])
rows = list(csv.DictReader(open("profiles/%s_kernel_stats.csv" % tag)))
side = ("pair_counts_kernel", "gram_kernel", "void prep1_stats_kernel<true>")
calls = max(int(r["Calls"]) for r in rows if r["Name"].startswith("pack_tables"))
tot = s_side = 0.0
for r in rows:
    name = r["Name"].split("(")[0]
    if name.startswith(("__amd", "void at::")): continue
    per = float(r["TotalDurationNs"]) / 1e3 / calls
    tot += per
    if name in side: s_side += per
ev = d["roofline"]["gpu_ms_per_step_hip_events"] * 1e3
open("profiles/%s_kernel_summary.txt" % tag, "a").write(
    "\nmain-stream launches: %.1f us per step (side-stream moment chain %.1f us runs beside conv_pool);\n"
    "HIP-event time per step in the default bench run (profiles/%s_bench.json, "
    "roofline.gpu_ms_per_step_hip_events): %.0f us\n-> launch gaps on the main stream ~%.0f us per step "
    "(12 launches)\n" % (tot - s_side, s_side, tag, ev, ev - (tot - s_side)))
print(d["value"], d["ms_per_step"], d["with_optimizer"], d["cpu_baseline"]["value"], d["roofline"]["measured_GBps"])
