#!/usr/bin/env python3
"""Copy one tools/final_profile.sh run (gpurun_out/<run>/) into profiles/<tag>_* and append the
trace-vs-HIP-event reconciliation to the kernel summary.   python tools/install_profiles.py r02_k r02_final"""
import csv, glob, json, shutil, sys
run, tag = sys.argv[1], sys.argv[2]
S = "gpurun_out/" + run
def cp(pattern, dst):
    # the newest match: gpurun merges every call's files into the same local directory
    import os
    shutil.copy(max(glob.glob(S + "/" + pattern), key=os.path.getmtime), "profiles/%s_%s" % (tag, dst))
cp("bench.json", "bench.json"); cp("stats/*/*kernel_stats.csv", "kernel_stats.csv")
cp("kernel_summary.txt", "kernel_summary.txt")
for p, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE"), ("sq1", "SQ1"), ("sq2", "SQ2")):
    cp(p + "/*/*counter_collection.csv", "pmc_%s.csv" % name)
cp("pmc_sq_summary.txt", "pmc_SQ_summary.txt")
t = open(S + "/counters.json").read().replace(run + "_pmc", tag + "_pmc")
open("profiles/%s_counters.json" % tag, "w").write(t)
d = json.loads(open("profiles/%s_bench.json" % tag).read().strip().splitlines()[-1])
rows = list(csv.DictReader(open("profiles/%s_kernel_stats.csv" % tag)))
calls = max(int(r["Calls"]) for r in rows if r["Name"].startswith("pack_tables"))
tot, n = 0.0, 0
for r in rows:
    name = r["Name"].split("(")[0]
    if name.startswith(("__amd", "void at::")): continue
    tot += float(r["TotalDurationNs"]) / 1e3 / calls
    n += int(r["Calls"]) // calls
ev = d["roofline"]["gpu_ms_per_step_hip_events"] * 1e3
open("profiles/%s_kernel_summary.txt" % tag, "a").write(
    "\n%d launches per step, %.1f us of kernel time per step (rocprofv3 --kernel-trace --stats);\n"
    "HIP-event time per step in the default bench run (profiles/%s_bench.json, "
    "roofline.gpu_ms_per_step_hip_events): %.0f us\n-> launch gaps ~%.0f us per step\n" % (n, tot, tag, ev, ev - tot))
print(d["value"], d["ms_per_step"], d["with_optimizer"], d["cpu_baseline"]["value"], d["roofline"]["measured_GBps"])
