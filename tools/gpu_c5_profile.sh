#!/bin/bash
# C5 (VGA, D=64, full get_depth, batch 128): rate and kernel stats
mkdir -p gpurun_out
python tools/gpu_depth_profile.py 640 480 64 128
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o c5 -- python tools/gpu_depth_profile.py 640 480 64 128 > /tmp/prof.log 2>&1
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/c5_kernel_stats.csv; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-60s calls %4s avg_us %9.1f  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
else tail -20 /tmp/prof.log; fi
