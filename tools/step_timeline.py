#!/usr/bin/env python3
"""One step's kernel timeline from a rocprofv3 --kernel-trace -f csv run: every dispatch of the LAST step (the launches after the
last but one occurrence of the step's first kernel), with start offset, duration and the idle gap before it.
usage: step_timeline.py <dir with *_kernel_trace.csv> [first-kernel substring]"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
key = sys.argv[2] if len(sys.argv) > 2 else "CatArrayBatchedCopy"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
if len(idx) < 2:
    sys.exit("no two occurrences of %r" % key)
step = rows[idx[-2]:idx[-1]]
t0, prev_end, busy = int(step[0]["Start_Timestamp"]), int(step[0]["Start_Timestamp"]), 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void sk::(anonymous namespace)::", "").replace("void at::native::", "at::")
    print("%8.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name[:110]))
    prev_end = max(prev_end, e)
    busy += e - s
print("step: %d dispatches, %.1f us from first start to last end, %.1f us of kernel time" % (len(step), (prev_end - t0) / 1e3, busy / 1e3))
