#!/bin/bash
# round 6: k_cost occupancy target (launch_bounds waves per SIMD) on the new kernel: 5 / 6 (product) / 7 / 8
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_cost_occ.txt; : > $O
echo "# k_cost occupancy target: product = 6 waves per SIMD (66 VGPRs, 3 workgroups per CU), w8 = 64 VGPRs + 12 B scratch (4 per CU)" >> $O
bash tools/gpu_exp.sh "" w5 w7 w8 >> $O 2>&1
echo "== again (run-to-run)" >> $O
bash tools/gpu_exp.sh "" w8 >> $O 2>&1
echo "== parity w8" >> $O
CAMD_LIB=$PWD/calibrating_amd/lib/dbg_w8.so timeout 900 python -m pytest tests/test_gpu_sgbm.py -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -2 >> $O
cat $O
