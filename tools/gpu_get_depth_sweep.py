"""get_depth / get_depth_batch over image sizes, disparity counts and batch sizes (device-resident): ms per pair, so that
slow corners of the whole path (not only the matcher) stand out."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import calibrating_amd as ca
from calibrating_amd import synthetic
dev = torch.device("cuda", 0)

def t(fn, reps):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3

for (W, H) in ((640, 480), (1280, 720), (1920, 1080), (3840, 2160)):
    for D in (16, 64, 128, 256):
        if W - D < 200: continue
        stereo = ca.Stereo.load(synthetic.rig(W, H))
        cfg = dict(max_size=max(W, H), minDisparity=0, numDisparities=D, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
                   uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
        stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=3.5)
        i1, i2 = synthetic.scene_pair(9, W, H, 3)
        t1, t2 = torch.from_numpy(i1).to(dev), torch.from_numpy(i2).to(dev)
        line = "%4dx%-4d D=%3d  get_depth %.3f ms |" % (W, H, D, t(lambda: stereo.get_depth(t1, t2), 8))
        for nb in (2, 8, 32):
            if nb * W * H * D > 32 * 1920 * 1080 * 128: continue
            B1, B2 = t1[None].repeat(nb, 1, 1, 1), t2[None].repeat(nb, 1, 1, 1)
            line += "  batch %2d: %.3f ms/pair" % (nb, t(lambda: stereo.get_depth_batch(B1, B2), 3) / nb)
            del B1, B2
        print(line, flush=True)
        del stereo
        torch.cuda.empty_cache()
