#!/bin/bash
# engine clock while the bench kernels run: rocm-smi sampled every 0.25 s beside a loop of one kernel at a time (CAMD_OPT_PHASES)
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_clocks.txt; : > $O
cat > /tmp/phase_loop.py <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
import calibrating_amd as ca
from calibrating_amd import synthetic
ph = int(sys.argv[1])
dev = torch.device("cuda", 0)
cn = 3
m = ca.StereoSGBM_create(minDisparity=0, numDisparities=128, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10)
L, R = synthetic.rectified_batch_torch(1234, 64, 1080, 1920, 128, cn, dev)
out = torch.empty((64, 1080, 1920), dtype=torch.int16, device=dev)
m.set_option("path", 2); m.compute(L, R, out=out); torch.cuda.synchronize()
m.set_option("phases", ph)
t0 = time.time(); n = 0
while time.time() - t0 < 6.0:
    for _ in range(10): m.compute(L, R, out=out)
    torch.cuda.synchronize(); n += 10
print("phase %d: %.2f ms per launch" % (ph, (time.time() - t0) / n * 1e3), flush=True)
PY
for PH in 1 2 4; do
  python /tmp/phase_loop.py $PH > /tmp/loop_$PH.log 2>&1 &
  PID=$!
  sleep 3.5
  for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | head -2 | tr "\n" " "; echo; sleep 0.25; done > /tmp/clk_$PH.txt
  wait $PID
  echo "== $(grep phase /tmp/loop_$PH.log)" >> $O
  cat /tmp/clk_$PH.txt >> $O
done
echo "== idle" >> $O; rocm-smi --showclocks --showpower --showmaxpower 2>/dev/null | grep -i "sclk\|Power" | head -4 >> $O
rocm-smi --showpower 2>/dev/null | grep -i "power" | head -3 >> $O
cat $O
