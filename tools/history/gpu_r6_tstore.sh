#!/bin/bash
# round 6: k_cost's C stores through an LDS tile (whole runs of NW x 16 B per pixel): product = gray only, ts3 = gray + RGB, ts0 = off
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_tstore.txt; : > $O
echo "# k_cost C stores through an LDS tile: product = gray only (CAMD_COST_TSTORE=1), ts3 = gray + RGB, ts0 = straight from the registers (rounds 2-5)" >> $O
echo "== parity product" >> $O
timeout 1500 python -m pytest tests/test_gpu_sgbm.py tests/test_gpu_int16_regime.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -2 >> $O
echo "== parity ts3" >> $O
CAMD_LIB=$PWD/calibrating_amd/lib/dbg_ts3.so timeout 1500 python -m pytest tests/test_gpu_sgbm.py tests/test_gpu_int16_regime.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -2 >> $O
echo "== gray 1080p D=128" >> $O
bash tools/gpu_exp.sh "--channels 1" ts0 >> $O 2>&1
echo "== C4" >> $O
bash tools/gpu_exp.sh "--channels 1 --width 3840 --height 2160 --disparities 256 --batch 16" ts0 >> $O 2>&1
echo "== RGB (product = register stores for RGB)" >> $O
bash tools/gpu_exp.sh "" ts3 ts0 >> $O 2>&1
cat $O
