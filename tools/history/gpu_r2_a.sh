#!/bin/bash
# round-2 first GPU pass: parity suite, RCCL path with one forced rank, --gpus 2 refusal, default bench, SQ PMC
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -x --timeout 900 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
tail -c 1500 gpurun_out/pytest_gpu.log
CAMD_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --steps 5 --warmup 1 --no-also --no-cpu-baseline > gpurun_out/rccl_1rank.log 2>&1
echo "rccl exit: $?" >> gpurun_out/rccl_1rank.log
tail -c 2500 gpurun_out/rccl_1rank.log
timeout 300 python bench.py --gpus 2 > gpurun_out/gpus2_on_1gpu_box.log 2>&1
echo "gpus2 exit: $?" >> gpurun_out/gpus2_on_1gpu_box.log
cat gpurun_out/gpus2_on_1gpu_box.log
timeout 900 python bench.py --steps 20 --warmup 2 > gpurun_out/bench_r02_v0.json 2> gpurun_out/bench_r02_v0.err
echo "bench exit: $?"; tail -c 3000 gpurun_out/bench_r02_v0.json; tail -5 gpurun_out/bench_r02_v0.err
bash tools/gpu_pmc.sh r02_v0 2>&1 | tail -15
