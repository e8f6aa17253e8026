"""BASELINE configs[4] (VGA, D=64, whole get_depth_batch, speckle on) against the pairs per call, one batch at a time and
two in flight (bench.depth_path_rate): what a deep batch buys an image this small.  -> gpurun_out/c5_batch_sweep.json"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import calibrating_amd as ca
from calibrating_amd import synthetic
dev = torch.device("cuda", 0)
W, H, D = 640, 480, 64
P = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=8 * 3 * 25, P2=32 * 3 * 25, disp12MaxDiff=1, preFilterCap=0,
         uniquenessRatio=10, speckleWindowSize=100, speckleRange=2, mode=0)
pairs = [synthetic.scene_pair(100 + i, W, H, 3) for i in range(8)]
out = {}
for nb in (32, 64, 128, 192, 256, 384, 512):
    B1 = torch.from_numpy(np.stack([pairs[i % 8][0] for i in range(nb)])).to(dev)
    B2 = torch.from_numpy(np.stack([pairs[i % 8][1] for i in range(nb)])).to(dev)
    r = bench.depth_path_rate(ca, synthetic, P, B1, B2, [], W, H, D, 3, max_depth=3.5, reps=6)
    out[nb] = {k: r[k] for k in ("pairs_per_s", "single_stream_pairs_per_s", "frac")}
    print(nb, out[nb], flush=True)
    del B1, B2
    torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"workload": "640x480 RGB, D=64, block 5, LR check, speckle 100/2, whole get_depth_batch", "by_pairs_per_call": out},
          open(os.path.join(ROOT, "gpurun_out", "c5_batch_sweep.json"), "w"), indent=1)
