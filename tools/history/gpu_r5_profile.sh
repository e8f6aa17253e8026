#!/bin/bash
# round-5 evidence pass on ONE box, on the code in the tree (CAMD_GIT_SHA names it):
#   gpurun -- 'CAMD_GIT_SHA=<sha> bash tools/gpu_r5_profile.sh'
# the default bench line; rocprofv3 kernel stats (both modes); HBM traffic (FETCH / WRITE passes at the headline's 64 pairs
# per launch) and SQ counters of the same command; C4 kernel stats + SQ counters; the default plugin's batch; latencies.
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
timeout 900 python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err; echo "bench exit: $?"
bash tools/gpu_profile.sh r05 --no-also > gpurun_out/prof_r05.log 2>&1
cp gpurun_out/prof_r05/bench_kernel_stats.csv gpurun_out/r05_bench_kernel_stats.csv
bash tools/gpu_profile.sh r05_hh --no-also --mode hh > gpurun_out/prof_r05_hh.log 2>&1
cp gpurun_out/prof_r05_hh/bench_kernel_stats.csv gpurun_out/r05_hh_bench_kernel_stats.csv
BATCH=64 bash tools/gpu_pmc_traffic.sh b64 > gpurun_out/pmct_r05.log 2>&1; cp gpurun_out/pmc_traffic_b64.json gpurun_out/r05_pmc_traffic_b64.json
BATCH=64 bash tools/gpu_pmc_traffic.sh hh_b64 --mode hh > gpurun_out/pmct_r05_hh.log 2>&1; cp gpurun_out/pmc_traffic_hh_b64.json gpurun_out/r05_pmc_traffic_hh_b64.json
bash tools/gpu_pmc.sh r05 > gpurun_out/pmc_r05.log 2>&1; cp gpurun_out/pmc_sq_r05.json gpurun_out/r05_pmc_sq.json
bash tools/gpu_pmc.sh r05_hh --mode hh > gpurun_out/pmc_r05_hh.log 2>&1; cp gpurun_out/pmc_sq_r05_hh.json gpurun_out/r05_pmc_sq_hh.json
C4="--width 3840 --height 2160 --disparities 256 --channels 1 --no-also --no-cpu-baseline"
bash tools/gpu_profile.sh c4 $C4 --batch 16 > gpurun_out/c4_profile.log 2>&1; cp gpurun_out/prof_c4/bench_kernel_stats.csv gpurun_out/r05_c4_kernel_stats.csv
bash tools/gpu_pmc.sh c4 $C4 > gpurun_out/c4_pmc.log 2>&1; cp gpurun_out/pmc_sq_c4.json gpurun_out/r05_c4_pmc_sq.json
bash tools/gpu_r5_default_batch_profile.sh > gpurun_out/default_batch_profile.log 2>&1; tail -30 gpurun_out/default_batch_profile.log | head -12
timeout 600 python tools/gpu_default_matcher_latency.py > gpurun_out/default_matcher.log 2>&1; tail -16 gpurun_out/default_matcher.log
timeout 600 python tools/gpu_latency.py > gpurun_out/latency.log 2>&1; tail -3 gpurun_out/latency.log
timeout 600 python tools/gpu_numpy_latency.py > gpurun_out/numpy_latency.log 2>&1; cp gpurun_out/numpy_latency.json gpurun_out/r05_numpy_latency.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_bench.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"]["dominant_kernel"]["stage"], d["also"]["get_depth_batch_pairs_per_s"], d["also"]["c5"]["pairs_per_s"], d["also"]["c4"]["pairs_per_s"], d["cpu_baseline"]["value"])
PY
