#!/bin/bash
# round 6: S shifted against C by a few byte offsets (CAMD_S_OFFSET): do the C / S streams of a pass collide in HBM?
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_s_offset.txt; : > $O
echo "# hipEvent stage times, ms per 64 RGB pairs 1080p D=128, with the S volume shifted by CAMD_S_OFFSET bytes against its allocation" >> $O
for OFF in 0 256 1024 4096 16384 65536 262144 1048576 0; do
  echo "== CAMD_S_OFFSET=$OFF" >> $O
  CAMD_S_OFFSET=$OFF python tools/gpu_stage_probe.py --mode 0 2>&1 | grep -v amdgpu.ids | tail -1 >> $O
done
cat $O
