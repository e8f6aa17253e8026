"""Single-pair latency of the stages of Stereo.get_depth at 1080p (device-resident inputs)."""
import sys, os, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import calibrating_amd as ca
from calibrating_amd import synthetic, imgproc
dev = torch.device("cuda", 0)
def t(fn, reps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
res = {}
W, H = 1920, 1080
stereo = ca.Stereo.load(synthetic.rig(W, H))
i1, i2 = synthetic.scene_pair(9, W, H, 3)
t1, t2 = torch.from_numpy(i1).to(dev), torch.from_numpy(i2).to(dev)
tb = stereo._tables(dev)
res["rectify_x2_ms"] = t(lambda: stereo.rectify(t1, t2))
r1, r2 = stereo.rectify(t1, t2)
for mode in (0, 1):
    for path in (1, 2, 3):
        P = dict(minDisparity=0, numDisparities=128, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10, mode=mode)
        m = ca.StereoSGBM_create(**P); m.set_option("path", path)
        res["sgbm_mode%d_path%d_batch1_ms" % (mode, path)] = t(lambda: m.compute(r1, r2))
        for nb in (2, 4):
            L = r1[None].expand(nb, -1, -1, -1).contiguous(); R = r2[None].expand(nb, -1, -1, -1).contiguous()
            res["sgbm_mode%d_path%d_batch%d_ms_per_pair" % (mode, path, nb)] = t(lambda: m.compute(L, R), reps=5) / nb
        del m
cfg = dict(max_size=W, minDisparity=0, numDisparities=128, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10, speckleWindowSize=0)
stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=3.5)
res["get_depth_total_ms"] = t(lambda: stereo.get_depth(t1, t2))
d = stereo.get_depth(t1, t2)
res["unrectify_ms"] = t(lambda: stereo.unrectify_depth(d["rectify_depth"]))
res["undistort_ms"] = t(lambda: stereo.undistort_img(t1))
cfg["speckleWindowSize"] = 200; cfg["speckleRange"] = 2
stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=3.5)
res["get_depth_total_with_speckle_ms"] = t(lambda: stereo.get_depth(t1, t2))
print(json.dumps({k: round(v, 3) for k, v in res.items()}, indent=1))
json.dump(res, open("gpurun_out/latency.json", "w"), indent=1)
