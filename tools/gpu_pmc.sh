#!/bin/bash
# SQ / TCC PMC counters per kernel (own runs, kernel-trace only) for the bench command.
# Usage: gpurun -- 'bash tools/gpu_pmc.sh <tag> [bench args]'  ->  gpurun_out/pmc_sq_<tag>.json
TAG=${1:-pmc}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BATCH=16
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/raw$i -o p -- python $ROOT/bench.py --steps 1 --warmup 1 --in-flight 1 --batch $BATCH --no-also --no-cpu-baseline --no-pmc "$@" > $OUT/log$i.txt 2>&1
  find $OUT/raw$i -name "*counter_collection*" -exec cp {} $OUT/counters$i.csv \;
done
python - <<PY
import csv, collections, glob, json, re
res = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/counters*.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'camd::' not in k: continue
        k = re.sub(r"^void ", "", k.split('(')[0])
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k, v in agg.items():
        for c, x in v.items():
            res.setdefault(k, {})[c] = x / n[(k, c)]
for k, v in res.items():
    wc = v.get("SQ_WAVE_CYCLES"); busy = v.get("SQ_BUSY_CYCLES")
    if wc:
        v["derived"] = {
            "valu_active_frac_of_wave_cycles": v.get("SQ_ACTIVE_INST_VALU", 0) / wc,
            "wait_any_frac": v.get("SQ_WAIT_ANY", 0) / wc,
            "wait_inst_any_frac": v.get("SQ_WAIT_INST_ANY", 0) / wc,
            "active_inst_any_frac": v.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        }
    if v.get("SQ_LDS_IDX_ACTIVE"):
        v.setdefault("derived", {})["lds_bank_conflict_frac_of_lds_active"] = v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"]
json.dump({"command": "rocprofv3 --pmc <3 SQ counter sets, separate passes> --kernel-trace -- python bench.py --steps 1 --warmup 1 --batch $BATCH --no-also --no-cpu-baseline $*",
           "pairs_per_launch": $BATCH, "note": "per-launch averages; SQ_* cycle counters are in quad-cycles summed over SEs (MI355X_MICROARCH.md)",
           "kernels": res}, open("$ROOT/gpurun_out/pmc_sq_$TAG.json", "w"), indent=1)
for k, v in res.items():
    print(k[-70:], json.dumps(v.get("derived", {})))
PY
rm -rf $OUT/raw*
