"""ms per pair of the three aggregation paths against the number of pairs per call (1080p RGB, D=128)."""
import sys, os, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import calibrating_amd as ca
from calibrating_amd import synthetic
dev = torch.device("cuda", 0)
L, R = synthetic.rectified_batch_torch(1234, 64, 1080, 1920, 128, 3, dev)
res = {}
for mode in (0, 1):
    for path in (1, 2, 3):
        for nb in (1, 2, 4, 8, 16, 32, 64):
            P = dict(minDisparity=0, numDisparities=128, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10, mode=mode)
            m = ca.StereoSGBM_create(**P); m.set_option("path", path)
            out = torch.empty((nb, 1080, 1920), dtype=torch.int16, device=dev)
            try:
                for _ in range(2): m.compute(L[:nb], R[:nb], out=out)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                reps = 3 if nb >= 16 else 8
                for _ in range(reps): m.compute(L[:nb], R[:nb], out=out)
                torch.cuda.synchronize()
                res["mode%d_path%d_batch%d" % (mode, path, nb)] = round((time.perf_counter() - t0) / reps / nb * 1e3, 3)
            except Exception as e:
                res["mode%d_path%d_batch%d" % (mode, path, nb)] = str(e)[:40]
            del m, out
            torch.cuda.empty_cache()
for mode in (0, 1):
    print("mode", mode, "ms per pair; rows = path 1 scan / 2 band / 3 concurrent; cols = batch 1 2 4 8 16 32 64")
    for path in (1, 2, 3):
        print("  path", path, [res["mode%d_path%d_batch%d" % (mode, path, nb)] for nb in (1, 2, 4, 8, 16, 32, 64)])
json.dump(res, open("gpurun_out/batch_sweep.json", "w"), indent=1)
