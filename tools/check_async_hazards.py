#!/usr/bin/env python3
"""Scan the gfx950 ISA of the kernels that use hand-managed asynchronous loads (inline-asm global_load / ds_read whose
wait is a separate inline-asm s_waitcnt) for the one thing the compiler cannot know to avoid: an instruction that reads
or writes a destination register while the load is still in flight.

    python tools/check_async_hazards.py [file.hip ...]      exit status 1 if any hazard is found
    python tools/check_async_hazards.py --asm file.s ...    the same on ISA already emitted (csrc/Makefile: the BUILD GATE)

Model: a walk over each function in layout order, restarted with nothing pending after every unconditional branch.  A VMEM
load makes its destination VGPRs pending until an s_waitcnt whose vmcnt(N) leaves at most N loads outstanding (loads return
in order); a ds_read until lgkmcnt(0).  Compiler-generated loads are tracked too (harmless: the compiler waits before using
them).  Layout order is not control flow, so this is a lint, not a proof -- but every real instance seen so far (copies of a
pending register hoisted above the wait) shows up in it.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sigkernel_amd", "csrc")
DEFAULT = ["sk_wave_adj.hip", "sk_wave_deriv.hip", "sk_wave_fused.hip", "sk_wave_adj_fused.hip", "sk_wave_fused_mb.hip",
           "sk_wave_adj_fused_rbf.hip", "sk_wave_adj_fused_mb.hip", "sk_wave_deriv_fused.hip"]


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


def _blocks(body):
    """Basic blocks of one function: [(label or None, [instruction lines])] in layout order."""
    blocks, cur, label = [], [], None
    for raw in body.split("\n"):
        l = raw.strip()
        if not l or l.startswith(";"):
            continue
        m = re.match(r"(\.LBB\w+):", l)
        if m:
            if cur or label is not None:
                blocks.append((label, cur))
            cur, label = [], m.group(1)
            continue
        if l.startswith("."):
            continue
        l = l.split(";")[0].strip()
        cur.append(l)
        op = l.split()[0]
        if op.startswith("s_branch") or op.startswith("s_cbranch") or op == "s_endpgm":
            blocks.append((label, cur))
            cur, label = [], None
    if cur or label is not None:
        blocks.append((label, cur))
    return blocks


def _transfer(lines, vm, lgkm, hazards, name):
    """Walk one block from the pending state (vm: in-order list, lgkm: list); append hazards when `hazards` is a list."""
    vm, lgkm = list(vm), list(lgkm)
    for l in lines:
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
        if hazards is not None:
            touched = all_regs(l)
            for dest, text in vm + lgkm:
                if touched & dest:
                    hazards.append((name, text, l))
        is_lds_dma = (" lds" in l and op.startswith("buffer_load")) or op.startswith("global_load_lds")
        if (op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load")) and not is_lds_dma:
            vm.append((frozenset(regs(l.split()[1].rstrip(","))), l))
        elif is_lds_dma or op.startswith("global_store") or op.startswith("buffer_store") or "atomic" in op:
            vm.append((frozenset(), l))      # counts in vmcnt, no destination
        elif op.startswith("ds_read"):
            lgkm.append((frozenset(regs(l.split()[1].rstrip(","))), l))
    return vm, lgkm


def scan(asm_text):
    """A walk over every function in layout order, block by block.  Layout order is not control flow: after an unconditional
    branch (or s_endpgm) nothing falls through, so the walk restarts with nothing pending -- round 3: a loop whose latch is laid
    out between its pre-header and its header was otherwise flagged for the latch's read-ahead following the pre-header's."""
    hazards = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", asm_text, re.S | re.M):
        name = m.group(1)
        vm, lgkm = [], []
        for _, lines in _blocks(m.group(2)):
            vm, lgkm = _transfer(lines, vm, lgkm, hazards, name)
            last = lines[-1].split()[0] if lines else ""
            if last.startswith("s_branch") or last == "s_endpgm":
                vm, lgkm = [], []
    return hazards


def scan_pressure(asm_text):
    """Second rule (round 3): a kernel whose registers SPILL or overflow into AGPRs (more than 256 VGPRs) must not leave an LDS read
    in flight at all -- the allocator may spill or copy the read's destination right behind it, in code this lint's layout-order
    walk does not connect with the read (seen: NaN gradients from a variant with 150 spilled registers, last-bit run-to-run
    differences from one with 330).  In such a kernel every ds_read must be followed by nothing but more ds_reads up to its
    s_waitcnt lgkmcnt(0) (the blocking helpers: reads and wait in ONE asm statement)."""
    meta = {}
    for m in re.finditer(r"\.name:\s+(_Z\w+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", asm_text):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    out = []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", asm_text, re.S | re.M):
        name = m.group(1)
        vg, sp = meta.get(name, (0, 0))
        if vg <= 256 and sp == 0:
            continue
        lines = [l.split(";")[0].strip() for l in m.group(2).split("\n")]
        lines = [l for l in lines if l and not l.startswith(".") and not l.endswith(":")]
        pending = None
        for l in lines:
            op = l.split()[0]
            if op.startswith("ds_read"):
                pending = pending or l
            elif pending is not None:
                if op == "s_waitcnt" and "lgkmcnt(0)" in l:
                    pending = None
                else:
                    out.append((name, "%s  [%d VGPRs, %d spilled]" % (pending, vg, sp), l))
                    pending = None
    return out


def compile_to_asm(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-I" + CSRC, src,
           "-o", out]
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def main(argv):
    asm = bool(argv) and argv[0] == "--asm"      # the build gate (csrc/Makefile): the .s files -save-temps left next to the objects
    if asm:
        argv = argv[1:]
        if not argv:
            print("--asm needs the ISA files")
            return 2
    files = argv or [os.path.join(CSRC, f) for f in DEFAULT]
    bad = 0
    for f in files:
        text = open(f).read() if asm else compile_to_asm(f if os.path.isabs(f) or os.path.exists(f) else os.path.join(CSRC, f))
        hz = scan(text) + scan_pressure(text)
        print("%s: %d hazard(s)" % (os.path.basename(f), len(hz)))
        for name, load, use in hz[:10]:
            print("   %s\n      pending: %s\n      touched: %s" % (name, load, use))
        bad += len(hz)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
