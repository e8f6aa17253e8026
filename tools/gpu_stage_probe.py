"""Per-stage times of one batched compute for any mode / build: python tools/gpu_stage_probe.py [--mode N] [--batch B]
(CAMD_LIB selects a measurement build).  Prints the hipEvent stage table averaged over the timed calls."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import calibrating_amd as ca
from calibrating_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--mode", type=int, default=0)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--disparities", type=int, default=128)
ap.add_argument("--channels", type=int, default=3)
ap.add_argument("--block", type=int, default=5)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
cn = a.channels
m = ca.StereoSGBM_create(minDisparity=0, numDisparities=a.disparities, blockSize=a.block, P1=8 * cn * a.block ** 2,
                         P2=32 * cn * a.block ** 2, disp12MaxDiff=1, uniquenessRatio=10, mode=a.mode)
pairs = [synthetic.rectified_pair(seed=1234 + i, H=a.height, W=a.width, D=a.disparities, cn=cn) for i in range(min(a.batch, 8))]
L = torch.stack([torch.from_numpy(pairs[i % len(pairs)][0]) for i in range(a.batch)]).cuda()
R = torch.stack([torch.from_numpy(pairs[i % len(pairs)][1]) for i in range(a.batch)]).cuda()
m.set_option("path", 2)
m.set_profiling(True)
out = m.compute(L, R)
torch.cuda.synchronize()
acc = {}
for _ in range(a.reps):
    m.compute(L, R, out=out)
    torch.cuda.synchronize()
    for k, v in m.stage_times_ms().items():
        acc[k] = acc.get(k, 0.0) + v / a.reps
print("mode %d batch %d %dx%d D=%d cn=%d lib=%s" % (a.mode, a.batch, a.width, a.height, a.disparities, cn,
                                                  os.path.basename(os.environ.get("CAMD_LIB", "product"))))
print("  " + "  ".join("%s %.2f" % (k, v) for k, v in acc.items() if v > 0.005), " total %.2f ms" % sum(acc.values()))
