#!/bin/bash
# A long run of the four fuzzers on the code in the tree: gpurun -- 'CAMD_GIT_SHA=<sha> bash tools/gpu_fuzz_campaign.sh'
# (FIRST = first seed, NS / NR / NP = seeds of the SGBM, remap + speckle, pipeline fuzzers)
# -> gpurun_out/${TAG:-r05}_campaign_{sgbm,remap,speckle,pipeline}.log (one FUZZ line per seed: SHA, library hash, branch counts)
mkdir -p gpurun_out
timeout 7200 python tools/gpu_fuzz.py 1000 ${FIRST:-9000} ${NS:-30} > gpurun_out/${TAG:-r05}_campaign_sgbm.log 2>&1; echo "sgbm rc $?"; grep -c "^FUZZ" gpurun_out/${TAG:-r05}_campaign_sgbm.log; grep -c MISMATCH gpurun_out/${TAG:-r05}_campaign_sgbm.log
timeout 3600 python tools/gpu_fuzz_remap.py 1000 $((${FIRST:-9000} + 1000)) ${NR:-10} > gpurun_out/${TAG:-r05}_campaign_remap.log 2>&1; echo "remap rc $?"; grep -c MISMATCH gpurun_out/${TAG:-r05}_campaign_remap.log
timeout 3600 python tools/gpu_fuzz_speckle.py 1000 $((${FIRST:-9000} + 2000)) ${NR:-10} > gpurun_out/${TAG:-r05}_campaign_speckle.log 2>&1; echo "speckle rc $?"; grep -c MISMATCH gpurun_out/${TAG:-r05}_campaign_speckle.log
timeout 7200 python tools/gpu_fuzz_pipeline.py 300 $((${FIRST:-9000} + 3000)) ${NP:-10} > gpurun_out/${TAG:-r05}_campaign_pipeline.log 2>&1; echo "pipeline rc $?"; grep -c MISMATCH gpurun_out/${TAG:-r05}_campaign_pipeline.log
