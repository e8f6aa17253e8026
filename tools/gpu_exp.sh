#!/bin/bash
# A/B of library variants built by tools/build_dbg.sh: tools/gpu_exp.sh "<bench args>" name [name ...]
ARGS=$1; shift
run() { python bench.py --no-cpu-baseline --no-also --no-pmc --steps 20 --warmup 2 --in-flight 1 $ARGS "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('pairs/s %.1f  ms/step %.2f  stages %s' % (d['value'], d['ms_per_step'], {k: round(v['avg_ms_per_launch'],2) for k,v in r['kernels'].items()}))
"; }
echo "== product"; run
for V in "$@"; do echo "== $V"; run --lib calibrating_amd/lib/dbg_$V.so; done
