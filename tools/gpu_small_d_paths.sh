run() { python bench.py --no-cpu-baseline --no-also --steps 6 --warmup 2 --in-flight 1 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('   %8.1f pairs/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"; }
for cfg in "1920 1080 16 16" "1920 1080 16 32" "1920 1080 32 8" "1920 1080 32 16" "1920 1080 32 32" "640 480 16 16" "640 480 16 64" "640 480 16 128" "640 480 32 16" "640 480 32 64" "1280 720 16 32" "1280 720 32 16"; do set -- $cfg
  for P in 1 2; do echo -n "W=$1 H=$2 D=$3 batch=$4 path=$P"; run --width $1 --height $2 --disparities $3 --batch $4 --path $P; done; done
