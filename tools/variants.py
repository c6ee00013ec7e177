#!/usr/bin/env python3
"""(CPU) Every kernel instance of the build: VGPRs (arch + accum), SGPRs, scratch bytes, static LDS -- from the gfx950 ISA that
csrc/Makefile keeps in csrc/obj (-save-temps).  usage: variants.py [--csv] [substring ...]
With --reached <rocprofv3 kernel_stats.csv | dir of them | the HIP runtime's launch log reduced to "<count> ShaderName : <name>" lines>: a
column of launches per instance (tools/reach_sweep.py is the workload).  With --check <file>: fails when an instance's scratch bytes grew against the committed table (profiles/r05_variants.txt)."""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "sigkernel_amd", "csrc", "obj")

def norm(name):
    """demangled kernel name -> its instance name without the parameter list (the same for hipcc's listing and rocprofv3's tables)"""
    n = name.replace("void ", "").replace("sk::(anonymous namespace)::", "").replace("sk::", "")
    i = n.find(">(")
    return n[:i + 1] if i >= 0 else n.split("(")[0]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return out[:len(names)]
    except Exception:
        return names

def table():
    rows = []
    for f in sorted(glob.glob(os.path.join(OBJ, "*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
        unit = os.path.basename(f).split("-hip-")[0]
        txt = open(f).read()
        for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", txt, re.S):
            name, body = m.group(1), m.group(2)
            def val(key, d=0):
                mm = re.search(r"\.amdhsa_%s (\d+)" % key, body)
                return int(mm.group(1)) if mm else d
            rows.append(dict(unit=unit, name=name, vgpr=val("next_free_vgpr"), accum=val("accum_offset"), sgpr=val("next_free_sgpr"),
                             scratch=val("private_segment_fixed_size"), lds=val("group_segment_fixed_size")))
    for r, d in zip(rows, demangle([r["name"] for r in rows])):
        r["pretty"] = norm(d)
    return rows

def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    rows = table()
    if "--check" in sys.argv:
        ref = {}
        for ln in open(args[0]):
            p = ln.rstrip("\n").split("\t")
            if len(p) >= 6 and p[0] != "unit":
                ref[p[-1]] = int(p[4])
        bad = [(r["name"], ref[r["name"]], r["scratch"]) for r in rows if r["name"] in ref and r["scratch"] > ref[r["name"]]]
        new_spill = [(r["name"], 0, r["scratch"]) for r in rows if r["name"] not in ref and r["scratch"] > 0]
        for n, a, b in bad + new_spill:
            print("scratch grew: %s %d -> %d bytes" % (n, a, b))
        sys.exit(1 if bad or new_spill else 0)
    reached = None
    if "--reached" in sys.argv:      # a rocprofv3 kernel_stats.csv of tools/reach_sweep.py: which instances the public API's routes launch
        import csv
        f = sys.argv[sys.argv.index("--reached") + 1]
        args = [a for a in args if a != f]
        reached = {}
        for ff in sorted(glob.glob(os.path.join(f, "**", "*kernel_stats.csv"), recursive=True)) if os.path.isdir(f) else [f]:
            if ff.endswith(".csv"):
                for r in csv.DictReader(open(ff)):
                    n = norm(r["Name"])
                    reached[n] = reached.get(n, 0) + int(r["Calls"])
            else:        # tools/reach_sweep.py --out: "<count><TAB><device symbol>" (sk_launch_trace_dump); older: "<count> ShaderName : <name>"
                lines = open(ff).read().split("\n")
                syms = [ln.split("\t", 1) for ln in lines if re.match(r"\d+\t\S", ln)]
                for (c, _), d in zip(syms, demangle([sname.split(".kd")[0] for _, sname in syms])):
                    n = norm(d)
                    reached[n] = reached.get(n, 0) + int(c)
                for ln in lines:
                    m = re.match(r"\s*(\d+) ShaderName : (.*)", ln)
                    if m:
                        n = norm(m.group(2).strip())
                        reached[n] = reached.get(n, 0) + int(m.group(1))
    if args:
        rows = [r for r in rows if all(a in r["pretty"] or a in r["unit"] for a in args)]
    print("unit\tvgpr\taccum_off\tsgpr\tscratch\tlds\t%sinstance\tmangled" % ("launches\t" if reached is not None else ""))
    for r in rows:
        extra = ("%d\t" % reached.get(r["pretty"], 0)) if reached is not None else ""
        print("%s\t%d\t%d\t%d\t%d\t%d\t%s%s\t%s" % (r["unit"], r["vgpr"], r["accum"], r["sgpr"], r["scratch"], r["lds"], extra, r["pretty"], r["name"]))
    print("# %d kernel instances, %d with scratch%s" % (len(rows), sum(1 for r in rows if r["scratch"]),
          (", %d never launched by the sweep" % sum(1 for r in rows if not reached.get(r["pretty"]))) if reached is not None else ""), file=sys.stderr)

if __name__ == "__main__":
    main()
