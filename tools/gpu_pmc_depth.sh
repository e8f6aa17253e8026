#!/bin/bash
# SQ counters + HBM traffic per kernel of the WHOLE get_depth_batch (speckle on): 64 pairs of 1080p.
#   gpurun -- 'bash tools/gpu_pmc_depth.sh'  ->  gpurun_out/pmc_depth.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_depth
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
            "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/raw$i -o p -- python $ROOT/tools/gpu_depth_profile.py 1920 1080 128 64 > $OUT/log$i.txt 2>&1
  find $OUT/raw$i -name "*counter_collection*" -exec cp {} $OUT/counters$i.csv \;
  rm -rf $OUT/raw$i
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
    wc = v.get("SQ_WAVE_CYCLES")
    d = {}
    if wc:
        d["wait_any_frac"] = v.get("SQ_WAIT_ANY", 0) / wc
        d["valu_active_frac_of_wave_cycles"] = v.get("SQ_ACTIVE_INST_VALU", 0) / wc
    if "FETCH_SIZE" in v or "WRITE_SIZE" in v:
        d["hbm_bytes_per_launch"] = (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024
    if v.get("SQ_INSTS_VALU") and v.get("GRBM_GUI_ACTIVE"):
        # VALU issue share: wave-instructions / (256 CUs x kernel cycles); GRBM_GUI_ACTIVE sums the 8 XCDs
        d["valu_issue_frac"] = v["SQ_INSTS_VALU"] / (256.0 * v["GRBM_GUI_ACTIVE"] / 8)
    v["derived"] = d
json.dump({"command": "rocprofv3 --pmc <4 passes> --kernel-trace -- python tools/gpu_depth_profile.py 1920 1080 128 64",
           "pairs_per_launch": 64, "note": "per-launch averages over the script's 5 calls; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
           "kernels": res}, open("$ROOT/gpurun_out/pmc_depth.json", "w"), indent=1)
for k, v in res.items():
    if any(s in k for s in ("k_cc_", "k_remap", "k_median", "k_lrcheck", "k_disp", "k_unrect")):
        print("%-42s %s" % (k[-42:], json.dumps({a: round(b, 3) if b < 10 else int(b) for a, b in v["derived"].items()})))
PY
