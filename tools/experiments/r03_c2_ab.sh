for c in c2 c2big; do for r in 1 2 3; do for w in r03d new; do python tools/r03_ab.py --one $w $c 2>&1 | grep -v "Warn\|amdgpu.ids"; done; done; done
