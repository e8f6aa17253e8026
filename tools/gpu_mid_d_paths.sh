run() { python bench.py --no-cpu-baseline --no-also --steps 6 --warmup 2 --in-flight 1 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('   %8.1f pairs/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"; }
for cfg in "640 480 64 8" "640 480 64 16" "640 480 64 32" "1280 720 64 8" "1280 720 64 16" "1280 720 128 8" "1920 1080 64 8" "1920 1080 64 16"; do set -- $cfg
  for P in 1 2 3; do echo -n "W=$1 H=$2 D=$3 batch=$4 path=$P"; run --width $1 --height $2 --disparities $3 --batch $4 --path $P; done; done
