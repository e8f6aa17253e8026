#!/bin/bash
# Round 5: evidence for BASELINE configs[3] (4K gray, D=256): kernel stats, SQ counters, pairs per call / in flight.
# Usage: gpurun -- 'bash tools/gpu_r5_c4.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
C4="--width 3840 --height 2160 --disparities 256 --channels 1 --no-also --no-cpu-baseline"
cd $ROOT
bash tools/gpu_profile.sh c4 $C4 --batch 16 > gpurun_out/c4_profile.log 2>&1
cp gpurun_out/prof_c4/*kernel_stats*.csv gpurun_out/r05_c4_kernel_stats.csv 2>/dev/null
bash tools/gpu_pmc.sh c4 $C4 > gpurun_out/c4_pmc.log 2>&1
cp gpurun_out/pmc_sq_c4.json gpurun_out/r05_c4_pmc_sq.json 2>/dev/null
for CFG in "--batch 16 --in-flight 1" "--batch 24 --in-flight 1" "--batch 12 --in-flight 2" "--batch 8 --in-flight 3"; do
  echo "== $CFG"
  python bench.py $C4 $CFG --steps 6 --warmup 1 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print(json.dumps(dict(pairs_per_s=round(d['value'], 1), ms_per_step=round(d['ms_per_step'], 2), frac=round(r['frac'], 3),
                              kernels={k: round(v['avg_ms_per_launch'], 2) for k, v in r['kernels'].items()})))
"
done 2>&1 | tee gpurun_out/r05_c4_batching.txt
