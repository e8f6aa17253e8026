#!/bin/bash
run() { echo "== ${ENVV[*]} $*"; env "${ENVV[@]}" timeout 600 python bench.py --no-cpu-baseline --no-also --steps 3 --warmup 1 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('pairs/s %.1f  ms/step %.2f  stages %s' % (d['value'], d['ms_per_step'], {k: round(v,2) for k,v in r['stage_ms_per_step'].items()}))
"; }
timeout 900 python -m pytest tests/test_gpu_sgbm.py tests/test_gpu_edge_cases.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
ENVV=(A=1)
run --mode sgbm --batch 64
run --mode sgbm --batch 64 --channels 1
run --mode sgbm --batch 64 --block 11

