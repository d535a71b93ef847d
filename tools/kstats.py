#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as a short table (per-step microseconds)."""
import csv, sys, glob
path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None   # (a third argument, the bench line, is for install_profiles.py)
f = glob.glob(path + "/*/*kernel_stats.csv")[0] if not path.endswith(".csv") else path
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows:
    if r["Name"].startswith(("__amd", "void at::")):
        continue
    calls = int(r["Calls"]); avg = float(r["AverageNs"]) / 1e3
    per = float(r["TotalDurationNs"]) / 1e3 / steps if steps else avg
    tot += per
    print("%-46s calls=%-4d avg=%8.1f us  per-step=%8.1f us" % (r["Name"].split("(")[0][:46], calls, avg, per))
print("TOTAL per step: %.1f us" % tot)
