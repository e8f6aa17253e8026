#!/bin/bash
# kernel start / end timestamps of the co-resident cost | last-pass pipeline (CAMD_OPT_RESIDENT 3:1), RGB C stores through the LDS tile
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trc
CAMD_LIB=$ROOT/calibrating_amd/lib/dbg_ts3.so timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trc -o t -- python $ROOT/tools/gpu_r6_pipeline.py --resident ${1:-31} --only-resident --steps 8 --out r06_tmp.json > $ROOT/gpurun_out/resident_trace.log 2>&1
f=$(find /tmp/trc -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $ROOT/gpurun_out/r06_resident_trace.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'camd::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
def short(n):
    for k in ('k_cost_persist', 'k_cost', 'k_wta_init', 'k_lrcheck', 'k_median3', 'k_band_row_persist'):
        if k in n: return k
    if 'k_band' in n: return 'band first' if ', true, 0,' in n else 'band last'
    return n[:20]
print("# last 40 kernels of the co-resident pipeline: k_cost_persist(k+1) gated on the end of the first pass of batch k, beside k_band_row_persist(k)")
for r in rows[-40:]:
    print("%-20s q%-3s start %9.3f ms  end %9.3f ms  dur %7.3f" % (short(r['Kernel_Name']), r.get('Queue_Id', '?'), (int(r['Start_Timestamp']) - t0) / 1e6,
          (int(r['End_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6))
PY
grep pairs $ROOT/gpurun_out/resident_trace.log
tail -28 $ROOT/gpurun_out/r06_resident_trace.txt
