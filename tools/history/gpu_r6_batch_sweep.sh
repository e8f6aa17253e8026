export TMPDIR=/tmp
for cfg in "64 2" "96 2" "128 1" "96 1" "80 2"; do set -- $cfg
python bench.py --no-cpu-baseline --no-also --no-pmc --steps 16 --warmup 3 --batch $1 --in-flight $2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('batch $1 in-flight $2: pairs/s %.1f  ms/step %.2f' % (d['value'], d['ms_per_step']))"
done
