#!/usr/bin/env python3
"""(CPU) Instruction histogram of the hottest loop of a kernel in hipcc's gfx950 assembly listing.
usage: isa_hist.py file.s <kernel-name-substring> [--all]
The hot loop is taken as the longest backward-branch span inside the kernel (the macro-step loop of the wavefront kernels)."""
import collections, re, sys

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\S*:", l) and all(k in l for k in key.split(",")):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\S+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch\S*\s+(\.LBB\S+)|\s+s_branch\s+(\.LBB\S+)", l)
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt < i:
                loops.append((i - tgt, tgt, i))
    # the hot loop: the SMALLEST loop that still holds most of the kernel's fp64 arithmetic (--outer: the largest loop)
    def nf64(lo, hi):
        return sum(1 for l in body[lo:hi + 1] if re.match(r"\s+v_\S*f64", l))
    tot64 = nf64(0, len(body) - 1)
    loops.sort()
    if "--outer" in sys.argv or not loops:
        _, lo, hi = loops[-1] if loops else (0, 0, len(body) - 1)
    else:
        _, lo, hi = next((lp for lp in loops if nf64(lp[1], lp[2]) >= 0.6 * tot64), loops[-1])
    h = collections.Counter()
    for l in body[lo:hi + 1]:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if m and not l.strip().startswith((";", ".")):
            h[m.group(1)] += 1
    def cls(op):
        if op.startswith("v_") and "f64" in op or op in ("v_rcp_f64", "v_ldexp_f64", "v_rndne_f64"): return "valu_f64"
        if "dpp" in op: return "valu_dpp"
        if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): return "valu_lane"
        if op.startswith("v_cndmask"): return "valu_select"
        if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "valu_mov"
        if op.startswith("v_cmp"): return "valu_cmp"
        if op.startswith("v_"): return "valu_other"
        if op.startswith("ds_"): return "lds"
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
        if op.startswith("s_waitcnt"): return "waitcnt"
        if op.startswith("s_"): return "salu"
        return "other"
    c = collections.Counter()
    for op, n in h.items():
        c[cls(op)] += n
    # dpp moves appear as v_mov_b32_dpp: count them out of movs
    tot = sum(h.values())
    print("kernel %s: loop lines %d..%d, %d instructions" % (body[0][:90], lo, hi, tot))
    for k, n in c.most_common():
        print("  %-12s %5d" % (k, n))
    valu = sum(n for k, n in c.items() if k.startswith("valu"))
    print("  VALU total %d, of which f64 %d" % (valu, c["valu_f64"]))
    if "--all" in sys.argv:
        for op, n in h.most_common(60):
            print("     %-28s %4d" % (op, n))

main()
