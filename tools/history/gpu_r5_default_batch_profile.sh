#!/bin/bash
# kernel stats of the reference's default plugin (max_size 1000, D=218, block 11, speckle 200/2), 64 pairs of 1080p per
# get_depth_batch call: 6 calls (2 warm-up + 3 timed + 1 profiled) -> gpurun_out/r05_default_plugin_batch_kernel_stats.csv
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_db -o db -- python $ROOT/tools/gpu_default_batch.py > $ROOT/gpurun_out/default_batch.log 2>&1
f=$(find /tmp/prof_db -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $ROOT/gpurun_out/r05_default_plugin_batch_kernel_stats.csv
tail -2 $ROOT/gpurun_out/default_batch.log
python3 - "$ROOT/gpurun_out/r05_default_plugin_batch_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:26]:
    print("%-70s calls=%5s total_ms=%8.2f avg_us=%9.1f  %4.1f%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
