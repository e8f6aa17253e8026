"""Single-pair and 16-pair SGBM time over a grid of configurations; prints cells/us so that slow corners stand out."""
import sys, os, time, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import calibrating_amd as ca
from calibrating_amd import synthetic
dev = torch.device("cuda", 0)
rows = []
for (W, H), D, bs, cn, mode in itertools.product([(640, 480), (1280, 720), (1920, 1080)], [16, 64, 128, 218, 320], [3, 5, 11, 15],
                                                 [1, 3], [0, 1, 2, 3]):
    if W - D < 100 or (mode == 2 and bs > 11):
        continue
    if (cn == 1 and mode in (2, 3)) or (bs in (3, 15) and mode != 0):
        continue  # thin the grid
    P = dict(minDisparity=0, numDisparities=D, blockSize=bs, P1=8 * cn * bs * bs, P2=min(32 * cn * bs * bs, 15000), disp12MaxDiff=1,
             uniquenessRatio=10, mode=mode)
    for nb in (1, 16):
        try:
            L, R = synthetic.rectified_batch_torch(5, nb, H, W, min(D, 128), cn, dev)
            m = ca.StereoSGBM_create(**P)
            out = torch.empty((nb, H, W), dtype=torch.int16, device=dev)
            m.compute(L, R, out=out); torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3): m.compute(L, R, out=out)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
            npaths = {0: 5, 1: 8, 2: 3, 3: 4}[mode]
            cells = nb * H * (W - D) * D
            rows.append((cells * npaths / ms / 1e3, W, H, D, bs, cn, mode, nb, ms))
            del m, L, R, out
        except Exception as e:
            print("FAILED", W, H, D, bs, cn, mode, nb, str(e)[:100])
        torch.cuda.empty_cache()
rows.sort()
print("slowest (cell-paths per us):")
for r in rows[:25]:
    print("%8.0f  %dx%d D=%d bs=%d cn=%d mode=%d batch=%d  %.3f ms" % r)
print("slowest at 16 pairs per call:")
for r in [r for r in rows if r[7] == 16][:25]:
    print("%8.0f  %dx%d D=%d bs=%d cn=%d mode=%d batch=%d  %.3f ms" % r)
print("fastest:")
for r in rows[-5:]:
    print("%8.0f  %dx%d D=%d bs=%d cn=%d mode=%d batch=%d  %.3f ms" % r)
