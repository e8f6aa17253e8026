#!/bin/bash
# round-3 baseline pass: parity suite, smoke, default bench.  Usage: gpurun -- 'bash tools/gpu_r3_base.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_r03_base.json 2> gpurun_out/bench_r03_base.err; echo "bench exit: $?"
tail -c 1500 gpurun_out/bench_r03_base.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_r03_base.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"]["dominant_kernel"]["kernel"], d["also"], d["cpu_baseline"]["value"])
PY
