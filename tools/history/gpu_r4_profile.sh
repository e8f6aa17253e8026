#!/bin/bash
# round-4 evidence pass on ONE box: the default bench line, rocprofv3 kernel stats (both modes, the whole
# get_depth_batch at 1080p and at C5, the default plugin's batch), latencies, the C5 batch sweep and the four fuzzers on
# the code that is in the tree (CAMD_GIT_SHA names it).   gpurun -- 'CAMD_GIT_SHA=<sha> bash tools/gpu_r4_profile.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
ROOT=$PWD
timeout 900 python bench.py > gpurun_out/bench_r04.json 2> gpurun_out/bench_r04.err; echo "bench exit: $?"
bash tools/gpu_profile.sh r04 --no-also > gpurun_out/prof_r04.log 2>&1
bash tools/gpu_profile.sh r04_hh --no-also --mode hh > gpurun_out/prof_r04_hh.log 2>&1
bash tools/gpu_r4_depth_profile.sh > gpurun_out/depth_profile_r04.log 2>&1; grep "sum of\|pairs_per" gpurun_out/depth_profile_r04.log
( cd /tmp && rm -rf /tmp/pdb && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pdb -o d -- python $ROOT/tools/gpu_default_batch.py > /tmp/pdb.log 2>&1; grep pairs_per /tmp/pdb.log; f=$(find /tmp/pdb -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $ROOT/gpurun_out/r04_default_plugin_batch_kernel_stats.csv )
timeout 600 python tools/gpu_default_matcher_latency.py > gpurun_out/default_matcher.log 2>&1; tail -16 gpurun_out/default_matcher.log
timeout 600 python tools/gpu_latency.py > gpurun_out/latency.log 2>&1; tail -3 gpurun_out/latency.log
timeout 600 python tools/gpu_c5_batch_sweep.py > gpurun_out/c5_sweep.log 2>&1; tail -3 gpurun_out/c5_sweep.log
timeout 1500 python tools/gpu_fuzz.py 1000 7000 6 > gpurun_out/r04_fuzz_sgbm.log 2>&1; grep -c MISMATCH gpurun_out/r04_fuzz_sgbm.log; tail -1 gpurun_out/r04_fuzz_sgbm.log | cut -c1-220
timeout 900 python tools/gpu_fuzz_remap.py 1000 7100 3 > gpurun_out/r04_fuzz_remap.log 2>&1; tail -1 gpurun_out/r04_fuzz_remap.log | cut -c1-220
timeout 900 python tools/gpu_fuzz_speckle.py 1000 7200 3 > gpurun_out/r04_fuzz_speckle.log 2>&1; tail -1 gpurun_out/r04_fuzz_speckle.log | cut -c1-220
timeout 1500 python tools/gpu_fuzz_pipeline.py 300 7300 3 > gpurun_out/r04_fuzz_pipeline.log 2>&1; tail -1 gpurun_out/r04_fuzz_pipeline.log | cut -c1-260
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r04.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"]["dominant_kernel"]["stage"], d["also"]["get_depth_batch_pairs_per_s"], d["also"]["c5"]["pairs_per_s"], d["cpu_baseline"]["value"])
PY
