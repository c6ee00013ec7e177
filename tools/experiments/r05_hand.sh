#!/bin/bash
# the handed-down edge pointer (HAND) against the build before it: times and VALU instruction counts of the forward that keeps edges
cd ${GRAFT_REPO_ROOT:-/root/repo}
export SK_AB_BASE=r05pre
python tools/ab.py e:rbf:1024:64:64:4:2 e:lin:512:128:128:8:1 e:rbf:512:128:128:4:1 c4 2>&1 | grep median
cd /tmp && export TMPDIR=/tmp
for which in r05pre new; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR --kernel-include-regex "k_fwd_fused" -f csv -d /tmp/hand_$which -o pmc -- python $GRAFT_REPO_ROOT/tools/ab.py --one $which e:rbf:1024:64:64:4:2 > /dev/null 2>&1
  python - $which <<'P'
import csv,glob,sys,collections
w=sys.argv[1]; acc=collections.defaultdict(list)
for f in glob.glob('/tmp/hand_%s/**/*counter_collection.csv'%w, recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r['Kernel_Name'][:70], r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(w, k[0], k[1], 'n=%d avg=%.6g'%(len(v), sum(v)/len(v)))
P
done
