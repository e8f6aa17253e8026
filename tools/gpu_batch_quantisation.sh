for B in 26 40 52 64 70 78; do python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 2 --in-flight 1 --batch $B 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; nb=$B
print('batch %3d  pairs/s %.1f  ms/pair: ' % (nb, d['value']) + ' '.join('%s %.4f' % (k, v['avg_ms_per_launch']/nb) for k,v in r['kernels'].items()))
"; done
