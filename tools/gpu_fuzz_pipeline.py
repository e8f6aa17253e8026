"""Seeded fuzz of the whole Stereo.get_depth path against the oracle-composed pipeline (run by hand on a GPU box):
random rigs (rotation, baseline, distortion, focal lengths), xy_target / K_target, max_depth (and with it the
min_disparity translation), matcher parameters incl. max_size downsizing, single calls and batches."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: F401
import calibrating_amd as ca
import oracle
from calibrating_amd import synthetic
from test_gpu_pipeline import _oracle_get_depth, DEPTH_TOL

oracle.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
inexact = {}
bad = checked_full = checked_batch = checked_down = 0
for case in range(n):
    W, H = int(rng.choice([160, 200, 256, 320])), int(rng.choice([96, 120, 150, 200]))
    f = W * rng.uniform(0.6, 1.1)
    K1 = [[f, 0, W / 2 + rng.uniform(-6, 6)], [0, f * rng.uniform(0.98, 1.02), H / 2 + rng.uniform(-5, 5)], [0, 0, 1]]
    K2 = [[f * rng.uniform(0.97, 1.03), 0, W / 2 + rng.uniform(-6, 6)], [0, f * rng.uniform(0.97, 1.03), H / 2 + rng.uniform(-5, 5)], [0, 0, 1]]
    rig = dict(R=synthetic.rodrigues(rng.uniform(-0.04, 0.04, 3)).tolist(),
               t=[[-rng.uniform(0.05, 0.3)], [rng.uniform(-0.01, 0.01)], [rng.uniform(-0.01, 0.01)]],
               cam1=dict(K=K1, D=[(rng.uniform(-1, 1, 5) * [0.2, 0.08, 2e-3, 2e-3, 0.02]).tolist()], xy=[W, H], name="a"),
               cam2=dict(K=K2, D=[(rng.uniform(-1, 1, 5) * [0.2, 0.08, 2e-3, 2e-3, 0.02]).tolist()], xy=[W, H], name="b"))
    xy_target = [None, None, 0.75, (W + 16, H - 8)][int(rng.integers(0, 4))]
    K_target = float(rng.choice([1, 1, 0.8, 1.2]))
    try:
        stereo = ca.Stereo(ca.Cam.load(rig["cam1"]), ca.Cam.load(rig["cam2"]), xy_target=xy_target, K_target=K_target,
                           R=np.array(rig["R"]), t=np.array(rig["t"]))
    except Exception as e:  # (a degenerate random rig)
        print("case", case, "rig refused:", str(e)[:80]); continue
    Wt, Ht = stereo.xy
    D = int(rng.choice([16, 32, 48, 64]))
    bs = int(rng.choice([3, 5, 7, 11]))
    if Wt - D < 24:
        continue
    cfg = dict(max_size=int(rng.choice([max(Wt, Ht), max(Wt, Ht), int(max(Wt, Ht) * 0.7)])), minDisparity=int(rng.integers(0, 4)),
               numDisparities=D, blockSize=bs, P1=8 * 3 * bs * bs, P2=32 * 3 * bs * bs, disp12MaxDiff=int(rng.integers(0, 3)),
               uniquenessRatio=int(rng.integers(0, 15)), speckleWindowSize=int(rng.choice([0, 60])), speckleRange=2,
               mode=int(rng.choice([0, 1, 3])))
    max_depth = [None, 3.0, 8.0][int(rng.integers(0, 3))]
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=max_depth)
    img1, img2 = synthetic.scene_pair(case, W, H, 3)
    got = stereo.get_depth(img1, img2)
    downsized = cfg["max_size"] < max(Wt, Ht)
    problems = []
    if not downsized:  # the oracle composition restates the full-resolution branch
        checked_full += 1
        ref = _oracle_get_depth(oracle, stereo, {k: v for k, v in cfg.items() if k != "max_size"}, img1, img2)
        for k in ("rectify_img1", "rectify_img2", "undistort_img1", "disparity"):
            if not np.array_equal(got[k], ref[k]): problems.append(k)
        for k in ("rectify_depth", "unrectify_depth"):
            if not np.array_equal(got[k] == 0, ref[k] == 0) or np.abs(got[k] - ref[k]).max() > DEPTH_TOL: problems.append(k)
            inexact[k] = inexact.get(k, 0) + int(not np.array_equal(got[k], ref[k]))
    else:  # the downsizing matcher (stereo_matching.py:60-70) composed from oracle stages, then stereo_camera.py:510-513
        checked_down += 1
        shift = stereo.min_disparity if stereo.translation_rectify_img else 0
        r1 = oracle.remap_u8(img1, *stereo.undistort_rectify_map1, oracle.INTER_LANCZOS4)
        r2 = oracle.remap_u8(img2, *stereo.undistort_rectify_map2, oracle.INTER_LANCZOS4)
        if shift > 0:
            r2[:, shift:] = r2[:, :-shift].copy()
            r2[:, :shift] = 0
        ratio = min(cfg["max_size"] / max(Ht, Wt), 1)
        hw = (int(round(Ht * ratio)), int(round(Wt * ratio)))
        sp = {k: v for k, v in cfg.items() if k != "max_size"}
        sd = oracle.sgbm_compute(oracle.resize_linear(r1, hw), oracle.resize_linear(r2, hw), **sp).astype(np.float32).clip(0)
        sd[sd < cfg["minDisparity"] * 16] = 0
        disp = oracle.resize_linear(sd / np.float32(16.0), (Ht, Wt)) * Wt / hw[1]
        if stereo.translation_rectify_img:
            disp += stereo.min_disparity
        disp = stereo.rectify_valid_mask1 * disp
        depth = stereo.disparity_to_depth(disp)
        if not np.array_equal(got["disparity"], disp): problems.append("down:disparity")
        if not np.array_equal(got["rectify_depth"], depth): problems.append("down:rectify_depth")
    # batched form == per-call form, always (covers the downsizing branch too)
    i1b, i2b = synthetic.scene_pair(case + 1000, W, H, 3)
    checked_batch += 1
    gb = stereo.get_depth_batch(np.stack([i1b, img1]), np.stack([i2b, img2]))
    for k in got:
        if not np.array_equal(gb[k][1], got[k], equal_nan=True): problems.append("batch:" + k)
    if problems:
        bad += 1
        print("MISMATCH case", case, dict(W=W, H=H, target=(Wt, Ht), cfg=cfg, max_depth=max_depth), problems, flush=True)
print("cases", n, "against the oracle", checked_full, "downsizing branch against the oracle", checked_down, "batch vs call", checked_batch, "mismatches", bad, "| cases whose depth is within tolerance but not bit-identical:", inexact)
