#!/usr/bin/env python3
"""Scan the gfx950 ISA of the kernels that use hand-managed asynchronous loads (inline-asm global_load / ds_read whose
wait is a separate inline-asm s_waitcnt) for the one thing the compiler cannot know to avoid: an instruction that reads
or writes a destination register while the load is still in flight.

    python tools/check_async_hazards.py [file.hip ...]      exit status 1 if any hazard is found

Model: a linear walk over each function in layout order.  A VMEM load makes its destination VGPRs pending until an
s_waitcnt whose vmcnt(N) leaves at most N loads outstanding (loads return in order); a ds_read until lgkmcnt(0).
Compiler-generated loads are tracked too (harmless: the compiler waits before using them).  Layout order is not
control flow, so this is a lint, not a proof -- but every real instance seen so far (copies of a pending register
hoisted above the wait) shows up in it.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sigkernel_amd", "csrc")
DEFAULT = ["sk_wave_adj.hip", "sk_wave_deriv.hip", "sk_wave_fused.hip", "sk_wave_adj_fused.hip", "sk_wave_fused_mb.hip",
           "sk_wave_adj_fused_rbf.hip"]


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def all_regs(line):
    out = set()
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", line):
        out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", line):
        out.add(int(a))
    return out


def scan(asm_text):
    hazards = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", asm_text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        vm = []      # pending VMEM loads, oldest first: (dest regs, text)
        lgkm = []    # pending LDS reads
        for raw in body.split("\n"):
            l = raw.strip()
            if not l or l.startswith(";") or (l.startswith(".") and not l.startswith(".LBB")):
                continue
            l = l.split(";")[0].strip()
            op = l.split()[0]
            if op == "s_waitcnt":
                mv = re.search(r"vmcnt\((\d+)\)", l)
                if mv:
                    n = int(mv.group(1))
                    vm = vm[len(vm) - n:] if n < len(vm) else vm
                    if n == 0:
                        vm = []
                if "lgkmcnt(0)" in l:
                    lgkm = []
                continue
            touched = all_regs(l)
            for dest, text in vm + lgkm:
                if touched & dest:
                    hazards.append((name, text, l))
            is_lds_dma = (" lds" in l and op.startswith("buffer_load")) or op.startswith("global_load_lds")
            if (op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load")) and not is_lds_dma:
                vm.append((regs(l.split()[1].rstrip(",")), l))
            elif is_lds_dma or op.startswith("global_store") or op.startswith("buffer_store") or "atomic" in op:
                vm.append((set(), l))      # counts in vmcnt, no destination
            elif op.startswith("ds_read"):
                lgkm.append((regs(l.split()[1].rstrip(",")), l))
    return hazards


def compile_to_asm(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-I" + CSRC, src,
           "-o", out]
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def main(argv):
    files = argv or [os.path.join(CSRC, f) for f in DEFAULT]
    bad = 0
    for f in files:
        hz = scan(compile_to_asm(f))
        print("%s: %d hazard(s)" % (os.path.basename(f), len(hz)))
        for name, load, use in hz[:10]:
            print("   %s\n      pending: %s\n      touched: %s" % (name, load, use))
        bad += len(hz)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
