#!/bin/bash
# PMC counters (own runs, kernel-trace only) for the bench command.  Usage: gpu_pmc.sh <tag> [bench args]
TAG=${1:-pmc}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/raw$i -o p -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $OUT/log$i.txt 2>&1
  find $OUT/raw$i -name "*counter_collection*" -exec cp {} $OUT/counters$i.csv \;
done
python - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/counters*.csv")):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r.get('Kernel_Name','')
        if 'camd::' not in k: continue
        k=k.split('(')[0][-40:]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[(k,r['Counter_Name'])]+=1
    for k,v in agg.items():
        print(f.split('/')[-1],k,{c:'%.4g'%(x/n[(k,c)]) for c,x in v.items()})
PY
rm -rf $OUT/raw*
