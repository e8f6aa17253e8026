#!/bin/bash
# round 6: k_cost's image-row fetches and C stores as buffer operations (row address in the descriptor = scalar arithmetic,
# lane part = 32-bit offset) vs global loads / stores from per-lane 64-bit pointers ("before")
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_bufaddr.txt; : > $O
echo "# k_cost: buffer loads / stores with the row in the descriptor (product) vs per-lane 64-bit pointers (before)" >> $O
echo "== parity product" >> $O
timeout 1500 python -m pytest tests/test_gpu_sgbm.py tests/test_gpu_int16_regime.py tests/test_gpu_edge_cases.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -2 >> $O
for i in 1 2; do
echo "== RGB" >> $O; bash tools/gpu_exp.sh "" before >> $O 2>&1
done
echo "== gray" >> $O; bash tools/gpu_exp.sh "--channels 1" before >> $O 2>&1
echo "== two in flight" >> $O
for V in "" before; do
  L=""; [ -n "$V" ] && L="--lib calibrating_amd/lib/dbg_$V.so"
  echo "-- ${V:-product}" >> $O
  python bench.py --no-cpu-baseline --no-also --no-pmc --steps 30 --warmup 3 $L 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('pairs/s %.1f  ms/step %.2f' % (d['value'], d['ms_per_step']))" >> $O
done
cat $O
