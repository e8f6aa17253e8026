#!/bin/bash
# A long run of the four fuzzers on the code in the tree: gpurun -- 'CAMD_GIT_SHA=<sha> bash tools/gpu_fuzz_campaign.sh'
# -> gpurun_out/r04_campaign_{sgbm,remap,speckle,pipeline}.log (one FUZZ line per seed: SHA, library hash, branch counts)
mkdir -p gpurun_out
timeout 2400 python tools/gpu_fuzz.py 1000 9000 30 > gpurun_out/r04_campaign_sgbm.log 2>&1; echo "sgbm rc $?"; grep -c "^FUZZ" gpurun_out/r04_campaign_sgbm.log; grep -c MISMATCH gpurun_out/r04_campaign_sgbm.log
timeout 1200 python tools/gpu_fuzz_remap.py 1000 9100 10 > gpurun_out/r04_campaign_remap.log 2>&1; echo "remap rc $?"; grep -c MISMATCH gpurun_out/r04_campaign_remap.log
timeout 1200 python tools/gpu_fuzz_speckle.py 1000 9200 10 > gpurun_out/r04_campaign_speckle.log 2>&1; echo "speckle rc $?"; grep -c MISMATCH gpurun_out/r04_campaign_speckle.log
timeout 2400 python tools/gpu_fuzz_pipeline.py 300 9300 10 > gpurun_out/r04_campaign_pipeline.log 2>&1; echo "pipeline rc $?"; grep -c MISMATCH gpurun_out/r04_campaign_pipeline.log
