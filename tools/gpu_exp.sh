#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_resize.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for cfg in "1920 1080 128 16" "640 480 64 128"; do
  rm -rf /tmp/dp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp -o p -- python $GRAFT_REPO_ROOT/tools/gpu_depth_profile.py $cfg 2>&1 | grep "get_depth_batch"
  python - <<PY
import csv,glob
f=glob.glob("/tmp/dp/**/*kernel_stats*.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:9]:
    print("  %-60s calls %5s total_ms %8.2f  %5.1f%%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["TotalDurationNs"])/1e6, 100*float(r["TotalDurationNs"])/tot))
PY
done
