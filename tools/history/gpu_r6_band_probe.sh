#!/bin/bash
# What bounds the first band pass?  (a) half the arithmetic at the same traffic: MODE_HH4 = {->, v} only;
# (b) the same arithmetic without the V of S stores; (c) without the per-step barrier.  hipEvent stage times, 64 RGB pairs.
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_band_probe.txt; : > $O
L=$PWD/calibrating_amd/lib
for M in 0 1 3; do python tools/gpu_stage_probe.py --mode $M >> $O 2>&1; done
for V in unmerged nostore nostore_u nobar; do
  for M in 0 3; do CAMD_LIB=$L/dbg_$V.so python tools/gpu_stage_probe.py --mode $M >> $O 2>&1; done
done
cat $O
