#!/bin/bash
run() { echo "== $V $*"; CAMD_LIB=$LIBP timeout 600 python bench.py --no-cpu-baseline --no-also --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('pairs/s %.1f  ms/step %.2f  stages %s' % (d['value'], d['ms_per_step'], {k: round(v,2) for k,v in r['stage_ms_per_step'].items()}))
"; }
for V in vfnosub vfnocst vfnone; do
  LIBP=$PWD/calibrating_amd/lib/dbg_$V.so
  run --mode sgbm --batch 64
done
