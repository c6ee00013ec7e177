#!/bin/bash
for cfg in c2 c2big mmd32 mmd64 c4fwd "e:rbf:1024:64:64:4:2" c5; do
  for rnd in 1 2; do
    for b in r06a r06noshift new; do SK_AB_BASE=$b python tools/ab.py --one $b "$cfg" 2>&1 | grep -v amdgpu; done
  done
done
for b in r06a new; do SK_AB_BASE=$b python tools/ab.py --one $b c4 2>&1 | grep -v amdgpu; done
python bench.py --steps 20 --warmup 10 --no-extras --no-configs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('plain bench', r['ms_per_step'])"
python bench.py --force-dist --steps 20 --warmup 10 --no-extras --no-configs 2>/dev/null | python -c "import sys,json; r=json.loads([l for l in sys.stdin.read().split('\n') if l.startswith('{')][-1]); print('force-dist bench', r['ms_per_step'], r.get('strong_scaling',{}).get('ms_per_step'), r.get('collectives',{}).get('all_gather_ms'))"
