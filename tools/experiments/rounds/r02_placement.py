#!/usr/bin/env python3
"""Parse the per-wave records an instrumented build of k_fwd_fused prints (C3 launches): where the waves ran and when.
usage: r02_placement.py log [launch index]"""
import sys, re, collections
recs = []
for l in open(sys.argv[1]):
    m = re.match(r"W (\d+) hw (\d+) xcc (\d+) t0 (\d+) t1 (\d+) cyc (\d+)", l)
    if m: recs.append(tuple(int(x) for x in m.groups()))
n = max(r[0] for r in recs) + 1
k = int(sys.argv[2]) if len(sys.argv) > 2 else 0
recs = recs[k * n:(k + 1) * n]
assert len(set(r[0] for r in recs)) == n, "launches interleaved"
t00 = min(r[3] for r in recs)
def where(hw): return ((hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3)   # se, sh, cu, simd
per_cu = collections.Counter(); per_simd = collections.Counter()
for w, hw, xcc, t0, t1, cyc in recs:
    se, sh, cu, simd = where(hw)
    per_cu[(xcc, se, sh, cu)] += 1; per_simd[(xcc, se, sh, cu, simd)] += 1
print("waves", len(recs), "CUs used", len(per_cu), "waves per CU histogram", sorted(collections.Counter(per_cu.values()).items()))
print("waves per SIMD histogram", sorted(collections.Counter(per_simd.values()).items()))
q = lambda a, f: a[int(f * (len(a) - 1))]
for name, a in (("start", sorted(r[3] - t00 for r in recs)), ("end", sorted(r[4] - t00 for r in recs)), ("duration", sorted(r[4] - r[3] for r in recs)),
                ("shader cycles", sorted(r[5] for r in recs))):
    print("%-14s min / 10%% / 50%% / 90%% / max: %d %d %d %d %d" % (name, a[0], q(a, .1), q(a, .5), q(a, .9), a[-1]))
by = collections.defaultdict(list)
for w, hw, xcc, t0, t1, cyc in recs:
    se, sh, cu, simd = where(hw)
    by[(per_cu[(xcc, se, sh, cu)], per_simd[(xcc, se, sh, cu, simd)])].append((t1 - t0, cyc, t0 - t00))
for key in sorted(by):
    v = by[key]
    print("  waves on CU = %2d, on SIMD = %d: n %4d, mean duration %.0f ticks, %.0f shader cycles, mean start %.0f" %
          (key[0], key[1], len(v), sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v), sum(x[2] for x in v) / len(v)))
