#!/bin/bash
# round 6, k_cost levers: parity of the product and of the DL = 16 build, then the A/B of the four builds
#   r5cost = round-5 tail + staging, product = new tail + staging, dl16 = + 16 disparities per lane (RGB), dl16g = gray too
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_cost_ab.txt; : > $O
echo "== parity, product" >> $O
timeout 1500 python -m pytest tests/test_gpu_sgbm.py tests/test_gpu_int16_regime.py -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -3 >> $O
for V in dl16 dl16g; do
  echo "== parity, $V" >> $O
  CAMD_LIB=$PWD/calibrating_amd/lib/dbg_$V.so timeout 1500 python -m pytest tests/test_gpu_sgbm.py tests/test_gpu_int16_regime.py -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -3 >> $O
done
echo "== A/B RGB, one batch in flight" >> $O
bash tools/gpu_exp.sh "" r5cost dl16 >> $O 2>&1
echo "== A/B RGB, two in flight" >> $O
for V in "" r5cost dl16; do
  L=""; [ -n "$V" ] && L="--lib calibrating_amd/lib/dbg_$V.so"
  echo "-- ${V:-product}" >> $O
  python bench.py --no-cpu-baseline --no-also --steps 30 --warmup 3 $L 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('pairs/s %.1f  ms/step %.2f' % (d['value'], d['ms_per_step']))" >> $O
done
echo "== A/B gray" >> $O
bash tools/gpu_exp.sh "--channels 1" r5cost dl16g >> $O 2>&1
cat $O
