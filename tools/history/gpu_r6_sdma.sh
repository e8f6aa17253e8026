#!/bin/bash
# the ndarray-in / dict-of-ndarrays-out call with the runtime's copies on the SDMA engines (default) vs on shader blit kernels
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_sdma.txt; : > $O
for S in "" 0; do
  [ -n "$S" ] && export HSA_ENABLE_SDMA=$S || unset HSA_ENABLE_SDMA
  echo "== HSA_ENABLE_SDMA=${S:-default}" >> $O
  timeout 600 python tools/gpu_numpy_latency.py > /tmp/nl.log 2>&1
  python - >> $O <<'PY'
import json
d = json.load(open("gpurun_out/numpy_latency.json"))
print({k: round(v, 2) for k, v in d.items() if k.startswith("1080p") and isinstance(v, (int, float))})
PY
done
cat $O
