#!/usr/bin/env python3
"""Aggregate a rocprofv3 counter_collection.csv: mean per dispatch of each counter, per kernel."""
import csv, sys, glob, collections
for path in sys.argv[1:]:
    f = glob.glob(path + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:34]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in acc.values() for c in k})
    print("%-34s " % "kernel" + " ".join("%14s" % c.replace("SQ_", "")[:14] for c in counters))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", kv[1].get(counters[0], [0])))):
        if k.startswith(("__amd", "at::")): continue
        print("%-34s " % k + " ".join("%14.4g" % (sum(v[c]) / max(len(v[c]), 1)) if c in v else "%14s" % "-" for c in counters))
