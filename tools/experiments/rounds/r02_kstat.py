#!/usr/bin/env python3
"""Print calls / average ns of the kernels whose name contains one of the given substrings, from a rocprofv3 --stats directory.
usage: r02_kstat.py dir substr [substr ...]"""
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in sys.argv[2:]):
            n = r["Name"].replace("void sk::(anonymous namespace)::", "").split("(")[0][:60]
            print("   %-60s calls %4s avg %10.3f ms" % (n, r["Calls"], float(r["AverageNs"]) / 1e6))
