#!/bin/bash
# round 6, band pass: 8 compute waves with the helper duty merged into wave 0 (product) vs 7 + 1 (unmerged)
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_band_ab.txt; : > $O
echo "== parity, product (merged)" >> $O
timeout 1500 python -m pytest tests/test_gpu_sgbm.py tests/test_gpu_int16_regime.py tests/test_gpu_edge_cases.py tests/test_gpu_configs.py -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -3 >> $O
echo "== A/B RGB MODE_SGBM, one batch in flight" >> $O
bash tools/gpu_exp.sh "" unmerged >> $O 2>&1
echo "== A/B RGB MODE_HH, one batch in flight" >> $O
bash tools/gpu_exp.sh "--mode hh" unmerged >> $O 2>&1
echo "== two in flight" >> $O
for V in "" unmerged; do
  L=""; [ -n "$V" ] && L="--lib calibrating_amd/lib/dbg_$V.so"
  for M in sgbm hh; do
  echo "-- ${V:-product} $M" >> $O
  python bench.py --no-cpu-baseline --no-also --steps 30 --warmup 3 --mode $M $L 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('pairs/s %.1f  ms/step %.2f' % (d['value'], d['ms_per_step']))" >> $O
  done
done
cat $O
