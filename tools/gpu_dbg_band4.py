import sys, numpy as np
sys.path.insert(0, "/root/repo")
import oracle
from calibrating_amd import StereoSGBM_create, synthetic
for (H, W, D, cn, mode) in ((48, 200, 128, 1, 0), (48, 200, 128, 1, 1), (20, 200, 128, 1, 0), (60, 150, 64, 1, 1)):
    left, right = synthetic.rectified_pair(seed=11 + D, H=H, W=W, D=D, cn=cn)
    p = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=8 * cn * 25, P2=32 * cn * 25, disp12MaxDiff=1, uniquenessRatio=10, mode=mode)
    m = StereoSGBM_create(**p); m.set_option("path", 2).set_option("keep_S", 1)
    got = m.compute(left, right)
    S = m.debug_volume("S").cpu().numpy().astype(int); Sr = oracle.sgbm_aggregated(left, right, **p).astype(int)
    bad = (S != Sr)
    print(H, W, D, "mode", mode, "S diff cells", bad.sum(), "of", bad.size, "final diff", (got != oracle.sgbm_compute(left, right, **p)).sum())
    if bad.any():
        rows = np.where(bad.any(axis=(1, 2)))[0]; cols = np.where(bad.any(axis=(0, 2)))[0]
        print(" rows", rows[:40].tolist(), "cols", cols[:20].tolist(), "...", cols[-5:].tolist())
        y, x = np.argwhere(bad.any(axis=2))[0]
        print(" first bad pixel", y, x, "got", S[y, x, :8].tolist(), "ref", Sr[y, x, :8].tolist(), "diff", (S[y, x] - Sr[y, x])[:16].tolist())
