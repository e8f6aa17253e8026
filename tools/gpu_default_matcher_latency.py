"""The reference's default plugin configuration (SemiGlobalBlockMatching({}): max_size=1000, numDisparities=218,
blockSize=11, speckle filter on) on 1080p input, one pair per call: the call an unmodified user of the reference makes."""
import sys, os, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import calibrating_amd as ca
from calibrating_amd import synthetic
W, H = 1920, 1080
stereo = ca.Stereo.load(synthetic.rig(W, H))
sm = ca.SemiGlobalBlockMatching({})
stereo.set_stereo_matching(sm, max_depth=3.5)
i1, i2 = synthetic.scene_pair(9, W, H, 3)
res = {}
for _ in range(3): out = stereo.get_depth(i1, i2)
t0 = time.perf_counter()
for _ in range(20): out = stereo.get_depth(i1, i2)
res["get_depth_numpy_in_out_ms"] = (time.perf_counter() - t0) / 20 * 1e3
t1, t2 = torch.from_numpy(i1).cuda(), torch.from_numpy(i2).cuda()
for _ in range(3): stereo.get_depth(t1, t2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): stereo.get_depth(t1, t2)
torch.cuda.synchronize(); res["get_depth_device_resident_ms"] = (time.perf_counter() - t0) / 20 * 1e3
r1, r2 = stereo.rectify(t1, t2)
for _ in range(3): sm(r1, r2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): sm(r1, r2)
torch.cuda.synchronize(); res["matcher_call_ms"] = (time.perf_counter() - t0) / 20 * 1e3
m = sm.stereo_sgbm
m.set_profiling(True); sm(r1, r2); torch.cuda.synchronize()
res["sgbm_stages_ms"] = {k: round(v, 3) for k, v in m.stage_times_ms().items()}
# the same configuration through the batched form: 64 pairs of 1080p per call
nb = 64
pairs = [synthetic.scene_pair(100 + i, W, H, 3) for i in range(4)]
B1 = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(nb)])).cuda()
B2 = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(nb)])).cuda()
for _ in range(2): stereo.get_depth_batch(B1, B2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): stereo.get_depth_batch(B1, B2)
torch.cuda.synchronize(); res["get_depth_batch64_pairs_per_s"] = nb * 3 / (time.perf_counter() - t0)
print(json.dumps(res, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/default_matcher.json", "w"), indent=1)
