#!/bin/bash
# round 4: kernel stats of the whole get_depth_batch at 1080p (64 pairs, speckle on) and at C5 (VGA, D=64, 128 pairs)
#   gpurun -- 'bash tools/gpu_r4_depth_profile.sh'   ->  gpurun_out/r04_{get_depth_batch,c5}_kernel_stats.csv
mkdir -p gpurun_out
export TMPDIR=/tmp
here=$PWD
show() { python - "$1" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print("%-64s calls %4s avg_us %9.1f  %5s%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
cc = sum(float(r["AverageNs"]) for r in rows if r["Name"].startswith("camd::k_cc_") or "k_cc_" in r["Name"]) / 1e6
print("sum of k_cc_* average launch times: %.3f ms" % cc)
PY
}
for cfg in "get_depth_batch 1920 1080 128 64" "c5 640 480 64 128"; do
    set -- $cfg
    python tools/gpu_depth_profile.py $2 $3 $4 $5
    ( cd /tmp && rm -rf /tmp/prof_$1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o $1 -- python $here/tools/gpu_depth_profile.py $2 $3 $4 $5 > /tmp/prof_$1.log 2>&1 )
    f=$(find /tmp/prof_$1 -name "*kernel_stats.csv" | head -1)
    if [ -n "$f" ]; then cp "$f" gpurun_out/r04_$1_kernel_stats.csv; show "$f"; else tail -20 /tmp/prof_$1.log; fi
done
