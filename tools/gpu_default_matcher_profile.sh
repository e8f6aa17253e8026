#!/bin/bash
# kernel stats of the reference's default plugin configuration, 64 pairs of 1080p per get_depth_batch call
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dm -o dm -- python $ROOT/tools/gpu_default_matcher_latency.py > $ROOT/gpurun_out/default_matcher.log 2>&1
f=$(find /tmp/prof_dm -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $ROOT/gpurun_out/default_matcher_kernel_stats.csv
tail -22 $ROOT/gpurun_out/default_matcher.log
python3 - "$ROOT/gpurun_out/default_matcher_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    print("%-64s calls=%s total_ms=%.2f avg_us=%.1f" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
