export TMPDIR=/tmp
echo "# product = 3 workgroups per CU (LDS floor 41 KB), occ4 = 4 per CU, before = per-lane pointers (3 per CU at 66 VGPRs)"
for i in 1 2 3; do
for V in "" occ4 before; do
  L=""; [ -n "$V" ] && L="--lib calibrating_amd/lib/dbg_$V.so"
  python bench.py --no-cpu-baseline --no-also --no-pmc --steps 40 --warmup 5 $L 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('${V:-product}: pairs/s %.1f  ms/step %.2f' % (d['value'], d['ms_per_step']))"
done
done
bash tools/gpu_exp.sh "" occ4 before
