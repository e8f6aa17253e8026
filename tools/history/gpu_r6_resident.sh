#!/bin/bash
# round 6: co-resident persistent launches (CAMD_OPT_RESIDENT a:b) of k_cost(k+1) beside the last pass of batch k
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_resident.txt; : > $O
for V in "" prio0 w8p; do
  L=""; [ -n "$V" ] && export CAMD_LIB=$PWD/calibrating_amd/lib/dbg_$V.so || unset CAMD_LIB
  echo "== ${V:-product (row-pass waves at s_setprio 3)}" >> $O
  python tools/gpu_r6_pipeline.py --resident 21,31,22 --out r06_resident_${V:-product}.json 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
