#!/bin/bash
# round-6 evidence pass on ONE box, on the code in the tree:   gpurun -- 'bash tools/gpu_r6_profile.sh'
# the default bench line (with its in-run rocprofv3 counters); rocprofv3 kernel stats of the same command (both modes);
# HBM traffic (FETCH / WRITE passes at the headline's 64 pairs per launch) and SQ counters per kernel; C4 kernel stats.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "bench exit: $?"
bash tools/gpu_profile.sh r06 --no-also > gpurun_out/prof_r06.log 2>&1
cp gpurun_out/prof_r06/bench_kernel_stats.csv gpurun_out/r06_bench_kernel_stats.csv
bash tools/gpu_profile.sh r06_hh --no-also --mode hh > gpurun_out/prof_r06_hh.log 2>&1
cp gpurun_out/prof_r06_hh/bench_kernel_stats.csv gpurun_out/r06_hh_bench_kernel_stats.csv
BATCH=64 bash tools/gpu_pmc_traffic.sh b64 > gpurun_out/pmct_r06.log 2>&1; cp gpurun_out/pmc_traffic_b64.json gpurun_out/r06_pmc_traffic_b64.json
BATCH=64 bash tools/gpu_pmc_traffic.sh hh_b64 --mode hh > gpurun_out/pmct_r06_hh.log 2>&1; cp gpurun_out/pmc_traffic_hh_b64.json gpurun_out/r06_pmc_traffic_hh_b64.json
bash tools/gpu_pmc.sh r06 > gpurun_out/pmc_r06.log 2>&1; cp gpurun_out/pmc_sq_r06.json gpurun_out/r06_pmc_sq.json
bash tools/gpu_pmc.sh r06_hh --mode hh > gpurun_out/pmc_r06_hh.log 2>&1; cp gpurun_out/pmc_sq_r06_hh.json gpurun_out/r06_pmc_sq_hh.json
C4="--width 3840 --height 2160 --disparities 256 --channels 1 --no-also --no-cpu-baseline"
bash tools/gpu_profile.sh c4 $C4 --batch 16 > gpurun_out/c4_profile.log 2>&1; cp gpurun_out/prof_c4/bench_kernel_stats.csv gpurun_out/r06_c4_kernel_stats.csv
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench.json"))
r = d["roofline"]
print(d["value"], r["frac"], r["counters"], r["valu_floor_ms"], r["hbm_floor_ms"], r["dominant_kernel"]["stage"],
      d["also"]["hh"]["pairs_per_s"], d["also"]["get_depth_batch_pairs_per_s"], d["also"]["c5"]["pairs_per_s"], d["also"]["c4"]["pairs_per_s"], d["cpu_baseline"]["value"])
PY
