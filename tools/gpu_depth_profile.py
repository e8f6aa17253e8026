"""Kernel-level timing of the full get_depth path (rectify x2, SGBM, depth, unrectify, undistort), batched."""
import sys, time, json, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import calibrating_amd as ca
from calibrating_amd import synthetic
W, H, D, nb = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 128, 16)
dev = torch.device("cuda", 0)
stereo = ca.Stereo.load(synthetic.rig(W, H))
cfg = dict(max_size=max(W, H), minDisparity=0, numDisparities=D, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10,
           speckleWindowSize=100, speckleRange=2)
stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=3.5)
# input: unrelated random textures (the speckle filter's worst case; default, as in rounds 1-3) or, with a fifth
# argument "scene", rendered textured planes (a real stereo scene: smooth disparities)
if len(sys.argv) > 5 and sys.argv[5] == "scene":
    planes = [((0.3, 0.1, 1.0), 2.0), ((-0.2, 0.15, 1.0), 1.6), ((0.0, 0.0, 1.0), 2.5), ((0.1, -0.25, 1.0), 1.3)]
    pairs = [synthetic.render_plane_pair(synthetic.rig(W, H), n_, d_, seed=i)[:2] for i, (n_, d_) in enumerate(planes)]
else:
    pairs = [synthetic.scene_pair(100 + i, W, H, 3) for i in range(4)]
B1 = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(nb)])).to(dev)
B2 = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(nb)])).to(dev)
for _ in range(2): stereo.get_depth_batch(B1, B2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): stereo.get_depth_batch(B1, B2)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(json.dumps({"size": [W, H, D], "batch": nb, "get_depth_batch_pairs_per_s": nb / dt, "ms_per_pair": dt / nb * 1e3}))
