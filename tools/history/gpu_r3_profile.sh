#!/bin/bash
# round-3 evidence pass: default bench, rocprofv3 kernel stats (both modes + the whole get_depth_batch), PMC traffic
# (both modes), SQ counters (both modes), batch sweep, latency.   gpurun -- 'bash tools/gpu_r3_profile.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err; echo "bench exit: $?"
bash tools/gpu_profile.sh r03 --no-also > gpurun_out/prof_r03.log 2>&1
bash tools/gpu_profile.sh r03_hh --no-also --mode hh > gpurun_out/prof_r03_hh.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_depth -o depth -- python $OLDPWD/tools/gpu_depth_profile.py 1920 1080 128 64 > /tmp/prof_depth.log 2>&1 )
f=$(find /tmp/prof_depth -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03_get_depth_batch_kernel_stats.csv
bash tools/gpu_pmc_traffic.sh sgbm > gpurun_out/pmct_sgbm.log 2>&1; tail -8 gpurun_out/pmct_sgbm.log
bash tools/gpu_pmc_traffic.sh hh --mode hh > gpurun_out/pmct_hh.log 2>&1; tail -8 gpurun_out/pmct_hh.log
bash tools/gpu_pmc.sh r03 > gpurun_out/pmc_sq_r03.log 2>&1; tail -8 gpurun_out/pmc_sq_r03.log
bash tools/gpu_pmc.sh r03_hh --mode hh > gpurun_out/pmc_sq_r03_hh.log 2>&1; tail -8 gpurun_out/pmc_sq_r03_hh.log
timeout 900 python tools/gpu_batch_sweep.py > gpurun_out/batch_sweep.log 2>&1; tail -3 gpurun_out/batch_sweep.log
timeout 600 python tools/gpu_latency.py > gpurun_out/latency.log 2>&1; tail -3 gpurun_out/latency.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r03.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"]["dominant_kernel"], d["also"]["get_depth_batch_pairs_per_s"], d["cpu_baseline"]["value"])
PY
