#!/bin/bash
# PMC counters of the Lanczos remap kernel variants (tools/microtests/remap_bench.hip).  gpurun -- 'bash tools/gpu_remap_pmc.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/remap_pmc
cd $ROOT/tools/microtests && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -ffp-contract=off remap_bench.hip -o /tmp/rb || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA_|TCP_|TD_)[A-Z0-9_]+" | sort -u | head -150 > $ROOT/gpurun_out/remap_pmc/avail.txt
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAVES" \
            "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
            "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d /tmp/rp$i -o p -- /tmp/rb 16 > $ROOT/gpurun_out/remap_pmc/log$i.txt 2>&1
  find /tmp/rp$i -name "*counter_collection*" -exec cp {} $ROOT/gpurun_out/remap_pmc/counters$i.csv \;
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$ROOT/gpurun_out/remap_pmc/counters*.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'k_remap_f32<8, 3' not in k: continue
        k = k.split('(')[0][-22:] + " grid " + r.get("Grid_Size", "?")
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in sorted(agg):
        print(k, {c: round(sum(v)/len(v)) for c, v in agg[k].items()})
PY
