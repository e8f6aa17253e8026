import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle
from calibrating_amd import StereoSGBM_create
from test_gpu_edge_cases import _rand_pair
for (H, W, D) in ((30, 65, 64), (30, 66, 64), (5, 65, 64), (30, 33, 32), (30, 129, 128)):
    left, right = _rand_pair(H * 1000 + W, H, W, 1)
    for mode in (0, 1):
        p = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=200, P2=800, disp12MaxDiff=1, uniquenessRatio=10, mode=mode)
        ref = oracle.sgbm_compute(left, right, **p)
        rr = oracle.sgbm_compute(left, right, raw=True, **p)
        for path in (1, 2, 3):
            m = StereoSGBM_create(**p); m.set_option("path", path).set_option("keep_S", 1)
            got = m.compute(left, right)
            raw = m.debug_volume("raw").cpu().numpy()
            C = m.debug_volume("C").cpu().numpy(); Cr = oracle.sgbm_cost_volume(left, right, **p)
            msg = ""
            if path != 3:
                S = m.debug_volume("S").cpu().numpy(); Sr = oracle.sgbm_aggregated(left, right, **p)
                msg = "S diff %d" % (S != Sr).sum()
            print(H, W, D, "mode", mode, "path", path, "C diff", (C != Cr).sum(), msg, "raw diff", (raw != rr).sum(),
                  "final diff", (got != ref).sum(), "got", got[:, -1][:8].tolist(), "ref", ref[:, -1][:8].tolist())
