#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into a small text summary (kernel stats + PMC per kernel)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    for key in ("k_fwd_fused", "k_fwd_wave", "k_fwd_simple", "k_adj_simple", "k_adj_wave", "k_static_linear_adj", "k_static_rbf_adj",
                "k_static_linear", "k_static_rbf", "k_increments_adjoint", "k_increments"):
        if key in name:
            rest = name.split(key)[1]
            return key + (rest[:rest.index(">") + 1] if rest.startswith("<") and ">" in rest else "")
    return name.split("(")[0][:70]


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("%-72s %8s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
    for r in rows[:14]:
        print("%-72s %8s %14s %12s %8s" % (short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))

print()
print("== PMC (per dispatch averages, solver / increment kernels only) ==")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in acc:
            for c, vals in acc[k].items():
                print("%-40s %-28s n=%-4d avg=%.6g max=%.6g" % (k, c, len(vals), sum(vals) / len(vals), max(vals)))
if os.path.exists(os.path.join(out, "failed.txt")):
    print(open(os.path.join(out, "failed.txt")).read())
