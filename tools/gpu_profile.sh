#!/bin/bash
# rocprofv3 kernel-trace + stats of the bench command; summaries land in gpurun_out/prof_<tag>/
# Usage: gpurun -- 'bash tools/gpu_profile.sh <tag> [bench args]'
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --in-flight 1 --no-cpu-baseline --no-pmc "$@" > $OUT/bench_under_rocprof.log 2>&1
echo "rocprof exit: $?" >> $OUT/bench_under_rocprof.log
find $OUT/raw -name "*kernel_stats*" -exec cp {} $OUT/ \;
find $OUT/raw -name "*domain_stats*" -exec cp {} $OUT/ \;
ls -la $OUT $OUT/raw 2>/dev/null | head -30
for f in $OUT/*kernel_stats*.csv; do echo "== $f"; head -25 "$f"; done
tail -3 $OUT/bench_under_rocprof.log
rm -rf $OUT/raw
