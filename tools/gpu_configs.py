"""BASELINE.json configs beyond the headline: C4 (3840x2160 D=256) and C5 (640x480 D=64 full get_depth).
Writes gpurun_out/configs.json.  Timing only: parity of the same configs against the oracle lives in
tests/test_gpu_configs.py (the oracle is test infrastructure and is not imported here)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import calibrating_amd as ca
from calibrating_amd import synthetic

dev = torch.device("cuda", 0)
res = {}

def timeit(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps

# ---- C2 geometry, every aggregation mode of cv2.StereoSGBM (the bench line covers MODE_SGBM and MODE_HH in flight)
for mode, name in ((0, "sgbm"), (1, "hh"), (3, "hh4"), (2, "3way")):
    P = dict(minDisparity=0, numDisparities=128, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10, mode=mode)
    nb = 64
    L, R = synthetic.rectified_batch_torch(11, nb, 1080, 1920, 128, 3, dev)
    m = ca.StereoSGBM_create(**P)
    out = torch.empty((nb, 1080, 1920), dtype=torch.int16, device=dev)
    dt = timeit(lambda: m.compute(L, R, out=out), reps=3)
    m.status()
    res["C2_1080p_d128_rgb_%s_single_stream_pairs_per_s" % name] = nb / dt
    del m, L, R, out
    torch.cuda.empty_cache()

# ---- C4: 4K, D=256, gray, LDS-tiled aggregation on one GPU
for mode, name in ((0, "sgbm"), (1, "hh")):
    P = dict(minDisparity=0, numDisparities=256, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, uniquenessRatio=10, mode=mode)
    nb = 16  # (6 pairs: 172 pairs/s, 12: 196, 16: 208, 20: 200 -- the band wavefront of a pair fills slowly at 78 bands)
    L, R = synthetic.rectified_batch_torch(7, nb, 2160, 3840, 256, 1, dev)
    m = ca.StereoSGBM_create(**P)
    out = torch.empty((nb, 2160, 3840), dtype=torch.int16, device=dev)
    dt = timeit(lambda: m.compute(L, R, out=out), reps=2)
    m.status()
    res["C4_4k_d256_gray_%s_pairs_per_s" % name] = nb / dt
    del m, L, R, out
    torch.cuda.empty_cache()

# ---- C5: 640x480 RGB, D=64, full get_depth (rectify x2 + SGBM + depth + unrectify + undistort), LR check on
W, H = 640, 480
stereo = ca.Stereo.load(synthetic.rig(W, H))
cfg = dict(max_size=W, minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
           uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=3.5)
i1, i2 = synthetic.scene_pair(9, W, H, 3)
t1, t2 = torch.from_numpy(i1).to(dev), torch.from_numpy(i2).to(dev)
dt = timeit(lambda: stereo.get_depth(t1, t2), reps=20, warm=3)
res["C5_vga_d64_get_depth_device_resident_pairs_per_s"] = 1 / dt
dt = timeit(lambda: stereo.get_depth(i1, i2), reps=20, warm=3)
res["C5_vga_d64_get_depth_numpy_in_out_pairs_per_s"] = 1 / dt
# the same stages for 128 pairs per call (Stereo.get_depth_batch): every kernel launched once per batch
nb = 128
pairs = [synthetic.scene_pair(100 + i, W, H, 3) for i in range(8)]
B1 = torch.from_numpy(np.stack([pairs[i % 8][0] for i in range(nb)])).to(dev)
B2 = torch.from_numpy(np.stack([pairs[i % 8][1] for i in range(nb)])).to(dev)
dt = timeit(lambda: stereo.get_depth_batch(B1, B2), reps=5, warm=2)
res["C5_vga_d64_get_depth_batch128_device_resident_pairs_per_s"] = nb / dt
del B1, B2
# SGBM-only batch at VGA
P = {k: v for k, v in cfg.items() if k != "max_size"}
P["speckleWindowSize"] = 0
L, R = synthetic.rectified_batch_torch(3, 256, H, W, 64, 3, dev)
m = ca.StereoSGBM_create(**P)
out = torch.empty((256, H, W), dtype=torch.int16, device=dev)
dt = timeit(lambda: m.compute(L, R, out=out))
res["C5_vga_d64_sgbm_only_batch256_pairs_per_s"] = 256 / dt
print(json.dumps(res, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/configs.json", "w"), indent=1)
