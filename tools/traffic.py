#!/usr/bin/env python3
"""profiles/traffic.json from THIS round's PMC passes (tools/profile_round.sh): HBM bytes per launch of the kernels bench.py's
roofline objects name, as the microarchitecture guide prescribes -- TCC_EA0_RDREQ (128-byte requests x 128 B + the others x 64 B)
for reads, TCC_EA0_WRREQ (64-byte requests x 64 B + the others x 32 B) for writes, from passes of their own.
usage: traffic.py <gpurun_out/<tag>p> <gpurun_out/pmc_<tag>fwd> <tag>"""
import csv, glob, json, os, sys
from collections import defaultdict


def counters(d, kernel_substr):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def entry(c, pairs, source):
    rd128, rd = c.get("TCC_EA0_RDREQ_128B_sum", 0.0), c.get("TCC_EA0_RDREQ_sum", 0.0)
    wr64, wr = c.get("TCC_EA0_WRREQ_64B_sum", 0.0), c.get("TCC_EA0_WRREQ_sum", 0.0)
    read = rd128 * 128 + max(rd - rd128, 0.0) * 64
    write = wr64 * 64 + max(wr - wr64, 0.0) * 32
    return {"pairs_per_launch": pairs, "hbm_bytes_per_launch": read + write, "read_bytes": read, "write_bytes": write, "source": source}


def main():
    p3, pf = sys.argv[1], sys.argv[2]
    tag = sys.argv[3] if len(sys.argv) > 3 else "r04"
    rnd = "round " + tag.lstrip("r0")
    out = {}
    c = counters(os.path.join(p3, "pmc_c3"), "k_fwd_fused")
    if c:
        out["c3_fused"] = entry(c, 262144, rnd + ": rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum on "
                                "k_fwd_fused<double,1,false,false,false,0,8,4> (bench.py --config c3), tools/profile_round.sh -> profiles/" + tag + "p_rocprof_summary.txt")
    c = counters(os.path.join(p3, "pmc_c5"), "k_fwd_fused_mb")
    if c:
        out["c5_fused"] = entry(c, 65536, rnd + ": the same counters on k_fwd_fused_mb<float,2,true,1,16> (bench.py --config c5), profiles/" + tag + "p_rocprof_summary.txt: "
                                "path slabs re-fetched per band + band-boundary rows written through to L2 and read back")
    c = counters(pf, "k_fwd_wave")
    if c:
        out["c3"] = entry(c, 131072, rnd + ": the same counters on k_fwd_wave (tools/pmc_fwd.sh: 131072 pairs of 127 x 127, dyadic 1, fp64), "
                          "profiles/" + tag + "p_pmc_fwd_solver.txt")
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
