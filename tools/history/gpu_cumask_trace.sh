#!/bin/bash
# kernel start / end timestamps of the CU-mask pipeline (tools/gpu_cumask.py, N = 128): the cost kernel of batch k+1 on 128
# CUs beside the band passes of batch k on the other 128 -> gpurun_out/r05_cumask_trace.txt (the last variant run = "pipe 128")
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trc
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trc -o t -- python $ROOT/tools/gpu_cumask.py --splits 128 --steps 8 > $ROOT/gpurun_out/cumask_trace.log 2>&1
f=$(find /tmp/trc -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $ROOT/gpurun_out/r05_cumask_trace.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'camd::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
def short(n):
    for k in ('k_cost', 'k_wta_init', 'k_lrcheck', 'k_median3'):
        if k in n: return k
    if 'k_band' in n: return 'band1' if ', true, 0,' in n else 'band2'
    return n[:20]
print("# last 48 kernels of `tools/gpu_cumask.py --splits 128 --steps 8` = the variant 'pipe 128': k_cost of batch k+1 on a stream")
print("# masked to CUs 0..127, aggregation + post of batch k on a stream masked to CUs 128..255 (64 pairs of 1080p per launch)")
for r in rows[-48:]:
    print("%-10s q%-3s start %9.3f ms  end %9.3f ms  dur %7.3f" % (short(r['Kernel_Name']), r.get('Queue_Id', '?'), (int(r['Start_Timestamp']) - t0) / 1e6,
          (int(r['End_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
PY
grep pairs $ROOT/gpurun_out/cumask_trace.log
tail -30 $ROOT/gpurun_out/r05_cumask_trace.txt
