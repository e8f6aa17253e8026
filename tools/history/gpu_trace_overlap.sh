#!/bin/bash
# kernel start/end timestamps of the two-batches-in-flight bench (how much the kernels of the two streams overlap)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $ROOT/bench.py --steps 8 --warmup 2 --no-also --no-cpu-baseline "$@" > $ROOT/gpurun_out/trace_bench.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $ROOT/gpurun_out/trace_overlap.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'camd::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
def short(n):
    for k in ('k_cost', 'k_wta_init', 'k_lrcheck', 'k_median3'):
        if k in n: return k
    if 'k_band' in n: return 'band1' if ', true, 0,' in n else 'band2'
    return n[:20]
last = rows[-60:]
for r in last:
    print("%-10s q%-3s start %9.3f ms  end %9.3f ms  dur %7.3f" % (short(r['Kernel_Name']), r.get('Queue_Id', '?'), (int(r['Start_Timestamp']) - t0) / 1e6,
          (int(r['End_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
PY
tail -45 $ROOT/gpurun_out/trace_overlap.txt
