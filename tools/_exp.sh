for d in 0 1; do TUNE_WPC=8 python tools/tune_adj.py 131072 127 127 $d 2>&1 | grep default; done
TUNE_WPC=8 python tools/tune_adj.py 262144 63 63 2 2>&1 | grep default
python tools/time_split.py 131072 127 127 1
python tools/time_split.py 262144 63 63 2
python tools/time_split.py 131072 127 127 0
python tools/time_kgrad.py 256 256 128 8 1 linear 2>&1 | tail -1
python tools/time_kgrad.py 256 256 128 8 1 rbf 2>&1 | tail -1
python tools/time_mmd.py 512 128 8 1 linear 2>&1 | tail -2
python tools/time_mmd.py 512 128 8 1 rbf 2>&1 | tail -2
python tools/time_small.py 2>&1 | tail -8
