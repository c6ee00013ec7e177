#!/bin/bash
# GPU box: build k_adj_wave variants (prefetch distance / ring slots) and time forward+adjoint on the C3 and C4 tiles.
cd sigkernel_amd/csrc
for v in "1 1" "1 0" "2 0" "3 0"; do
  set -- $v
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DSK_ADJ_PF=$1 -DSK_ADJ_XSLOT=$2 -c sk_wave_adj.hip -o sk_wave_adj.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o ../libsigkernel_amd.so
  echo "== PF=$1 XSLOT=$2"
  (cd ../..; TUNE_WPC=8 python tools/tune_adj.py 131072 127 127 1 2>&1 | grep -v amdgpu | head -2; TUNE_WPC=8 python tools/tune_adj.py 262144 63 63 2 2>&1 | grep -v amdgpu | head -1; TUNE_WPC=4 python tools/tune_adj.py 131072 127 127 0 2>&1 | grep -v amdgpu | head -1)
done
