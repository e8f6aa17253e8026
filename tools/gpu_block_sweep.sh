for K in 1 3 5 7 9 11 13 15; do for CN in 1 3; do python bench.py --no-cpu-baseline --no-also --steps 5 --warmup 2 --in-flight 1 --batch 32 --block $K --channels $CN 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('block $K cn $CN  %.1f pairs/s' % d['value'], {k: round(v['avg_ms_per_launch'],2) for k,v in r['kernels'].items()})"; done; done
