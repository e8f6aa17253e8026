#!/bin/bash
# round 6: do RGB's full-line C stores (CAMD_COST_TSTORE=3) change what k_cost costs a kernel running beside it?
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_tstore_overlap.txt; : > $O
echo "# two batches in flight (bench default) and the co-resident cost|last pipeline, RGB C stores from registers (product) vs through the LDS tile (ts3)" >> $O
for i in 1 2; do
for V in "" ts3; do
  L=""; [ -n "$V" ] && L="--lib calibrating_amd/lib/dbg_$V.so"
  echo "-- bench, two in flight: ${V:-product}" >> $O
  python bench.py --no-cpu-baseline --no-also --no-pmc --steps 30 --warmup 3 $L 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('pairs/s %.1f  ms/step %.2f' % (d['value'], d['ms_per_step']))" >> $O
done
done
for V in "" ts3; do
  [ -n "$V" ] && export CAMD_LIB=$PWD/calibrating_amd/lib/dbg_$V.so || unset CAMD_LIB
  echo "== pipeline ${V:-product}" >> $O
  python tools/gpu_r6_pipeline.py --resident 21,31 --out r06_tmp.json 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
