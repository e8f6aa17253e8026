#!/bin/bash
# round-2 evidence pass: parity suite, default bench, rocprofv3 kernel stats (both modes), PMC traffic (both modes), SQ counters
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log; tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_r02.json 2> gpurun_out/bench_r02.err; echo "bench exit: $?"
bash tools/gpu_profile.sh r02 --no-also > gpurun_out/prof_r02.log 2>&1
bash tools/gpu_profile.sh r02_hh --no-also --mode hh > gpurun_out/prof_r02_hh.log 2>&1
bash tools/gpu_pmc_traffic.sh sgbm > gpurun_out/pmct_sgbm.log 2>&1; tail -8 gpurun_out/pmct_sgbm.log
bash tools/gpu_pmc_traffic.sh hh --mode hh > gpurun_out/pmct_hh.log 2>&1; tail -8 gpurun_out/pmct_hh.log
bash tools/gpu_pmc.sh r02 > gpurun_out/pmc_sq_r02.log 2>&1; tail -8 gpurun_out/pmc_sq_r02.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r02.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"]["dominant_kernel"]["kernel"], d["also"], d["cpu_baseline"]["value"])
PY
