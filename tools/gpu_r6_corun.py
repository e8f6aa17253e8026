#!/usr/bin/env python3
"""Round 6: do k_cost (VALU-bound) and the row-parallel last pass (HBM-bound) make progress side by side?
Stream A loops the cost kernel of one handle, stream B loops the last pass of another (CAMD_OPT_PHASES 1 / 4), with no
ordering between them; the wall time of both loops together is compared with each loop alone.  Also: cost beside the
first pass (phase 2), first beside last."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import calibrating_amd as ca
from calibrating_amd import synthetic

dev = torch.device("cuda", 0)
W, H, D, cn, nb = 1920, 1080, 128, 3, 64
P = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=8 * cn * 25, P2=32 * cn * 25, disp12MaxDiff=1,
         uniquenessRatio=10, mode=0)
L, R = synthetic.rectified_batch_torch(1234, nb, H, W, D, cn, dev)
ms = [ca.StereoSGBM_create(**P) for _ in range(2)]
outs = [torch.empty((nb, H, W), dtype=torch.int16, device=dev) for _ in range(2)]
for m, o in zip(ms, outs):
    m.set_option("path", 2)
    m.compute(L, R, out=o)
torch.cuda.synchronize()
s = [torch.cuda.Stream(), torch.cuda.Stream()]
NAMES = {1: "cost", 2: "first", 4: "last"}


def loop(i, phase, n):
    ms[i].set_option("phases", phase)
    with torch.cuda.stream(s[i]):
        for _ in range(n):
            ms[i].compute(L, R, out=outs[i])


def wall(jobs):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in jobs:
        loop(*j)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


res = []
for pa, pb in ((1, 4), (1, 2), (2, 4), (1, 1), (4, 4), (2, 2)):
    na = nb_ = 8
    for _ in range(2):
        ta = wall([(0, pa, na)])
        tb = wall([(1, pb, nb_)])
    # equalise the two loops' durations so that neither runs alone for long
    nb2 = max(1, round(nb_ * ta / tb))
    tb = wall([(1, pb, nb2)])
    both = wall([(0, pa, na), (1, pb, nb2)])
    r = dict(a=NAMES[pa], b=NAMES[pb], a_ms=ta, b_ms=tb, together_ms=both, sum_ms=ta + tb, max_ms=max(ta, tb),
             overlap_gain=(ta + tb - both) / min(ta, tb))
    res.append(r)
    print("%-5s x%d %.1f ms | %-5s x%d %.1f ms | together %.1f ms (sum %.1f, max %.1f): %.0f %% of the shorter loop hidden"
          % (NAMES[pa], na, ta, NAMES[pb], nb2, tb, both, ta + tb, max(ta, tb), 100 * r["overlap_gain"]), flush=True)
for m in ms:
    m.set_option("phases", 7)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r06_corun.json"), "w"), indent=1)
