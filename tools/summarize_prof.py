#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory: kernel-trace stats + PMC means per kernel."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k2u"
cands = glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True)
st = cands[0] if cands else ""
if st:
    print("== kernel-trace --stats (top 6 by total time, and every kernel matching '%s')" % pat)
    for i, r in enumerate(csv.DictReader(open(st))):
        if i < 6 or pat in r["Name"]:
            print("  %-60s calls=%s avg_ns=%.0f pct=%s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for kn, cs in agg.items():
        print("==", os.path.relpath(f, d).split(os.sep)[0], kn)
        for c, v in cs.items():
            print("   %-26s n=%d mean=%.5g" % (c, len(v), sum(v) / len(v)))
