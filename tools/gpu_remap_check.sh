#!/bin/bash
# remap change check: [tests,] the full get_depth_batch rate and its kernel stats
mkdir -p gpurun_out
[ "$1" = tests ] && python -m pytest tests -q -m gpu -x -k "remap or rectif or pipeline or golden or contract or stereo or config" 2>&1 | tail -5
python tools/gpu_depth_profile.py 1920 1080 128 64
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_depth -o depth -- python tools/gpu_depth_profile.py 1920 1080 128 64 > /tmp/prof.log 2>&1
f=$(find /tmp/prof_depth -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/depth_kernel_stats.csv; head -14 "$f" | cut -c1-60,200-; else tail -20 /tmp/prof.log; fi
