#!/bin/bash
# One 1080p pair per call (the reference's surface): kernel times (rocprofv3 --kernel-trace --stats) and HBM traffic
# (separate --pmc passes) of the path AUTO picks for it.  gpurun -- 'bash tools/gpu_single_pair_profile.sh [tag] [bench args]'
TAG=${1:-single}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o s -- python $ROOT/bench.py --batch 1 --steps 30 --warmup 3 --in-flight 1 --no-also --no-cpu-baseline "$@" > /tmp/prof_$TAG.log 2>&1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' /tmp/prof_$TAG.log | head -2
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $ROOT/gpurun_out/${TAG}_kernel_stats.csv && python - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-70s calls %4s avg_us %9.1f  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cd $ROOT && BATCH=1 bash tools/gpu_pmc_traffic.sh $TAG "$@" 2>&1 | tail -12
