"""BASELINE configs[4] (VGA, D=64, whole get_depth_batch, speckle on) against the number of batches in flight
(one Stereo + stream each) at a few pairs per call."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import calibrating_amd as ca
from calibrating_amd import synthetic
dev = torch.device("cuda", 0)
W, H, D = 640, 480, 64
P = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=8 * 3 * 25, P2=32 * 3 * 25, disp12MaxDiff=1, preFilterCap=0,
         uniquenessRatio=10, speckleWindowSize=100, speckleRange=2, mode=0, max_size=W)
pairs = [synthetic.scene_pair(100 + i, W, H, 3) for i in range(8)]
out = {}
for nb in (128, 256):
    B1 = torch.from_numpy(np.stack([pairs[i % 8][0] for i in range(nb)])).to(dev)
    B2 = torch.from_numpy(np.stack([pairs[i % 8][1] for i in range(nb)])).to(dev)
    for nfl in (1, 2, 3, 4):
        sts, streams = [], [torch.cuda.Stream() for _ in range(nfl)]
        for s in streams:
            st = ca.Stereo.load(synthetic.rig(W, H)); st.set_stereo_matching(ca.SemiGlobalBlockMatching(P), max_depth=3.5)
            with torch.cuda.stream(s): st.get_depth_batch(B1, B2)
            sts.append(st)
        torch.cuda.synchronize(); reps = 6; t0 = time.perf_counter()
        for k in range(nfl * reps):
            with torch.cuda.stream(streams[k % nfl]): sts[k % nfl].get_depth_batch(B1, B2)
        torch.cuda.synchronize()
        out["%d pairs per call, %d in flight" % (nb, nfl)] = nb * nfl * reps / (time.perf_counter() - t0)
        print(nb, nfl, out["%d pairs per call, %d in flight" % (nb, nfl)], flush=True)
        del sts
        torch.cuda.empty_cache()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c5_inflight.json"), "w"), indent=1)
