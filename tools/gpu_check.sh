#!/bin/bash
# One GPU-box pass: parity tests, smoke, short bench.  Usage: gpurun -- 'bash tools/gpu_check.sh [pytest-args]'
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m2 -E "gfx|Marketing" > gpurun_out/device.txt
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
tail -c 6000 gpurun_out/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench exit: $?" >> gpurun_out/bench.log
tail -c 3000 gpurun_out/bench.log
