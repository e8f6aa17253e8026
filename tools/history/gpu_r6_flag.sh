#!/bin/bash
# round 6: edge flags raised FLAG_DELAY steps late behind a counted s_waitcnt (product: 2) vs drained at once (flag0)
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_flag_delay.txt; : > $O
echo "# edge-record flags: product = raised 2 steps after the chunk's last column behind s_waitcnt vmcnt(2 x ops per step); flag1 = 1 step; flag0 = s_waitcnt vmcnt(0) at once (rounds 1-5)" >> $O
echo "== parity, product" >> $O
timeout 1500 python -m pytest tests/test_gpu_sgbm.py tests/test_gpu_int16_regime.py tests/test_gpu_edge_cases.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -q -x --timeout 900 -p no:cacheprovider 2>&1 | tail -3 >> $O
for M in sgbm hh; do
echo "== A/B RGB mode $M, one batch in flight" >> $O
bash tools/gpu_exp.sh "--mode $M" flag1 flag0 >> $O 2>&1
done
echo "== two in flight" >> $O
for V in "" flag0; do
  L=""; [ -n "$V" ] && L="--lib calibrating_amd/lib/dbg_$V.so"
  for M in sgbm hh; do
  echo "-- ${V:-product} $M" >> $O
  python bench.py --no-cpu-baseline --no-also --no-pmc --steps 30 --warmup 3 --mode $M $L 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('pairs/s %.1f  ms/step %.2f' % (d['value'], d['ms_per_step']))" >> $O
  done
done
cat $O
