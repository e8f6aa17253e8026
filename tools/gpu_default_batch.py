"""The reference's default plugin (SemiGlobalBlockMatching({}): max_size 1000, D=218, block 11, speckle 200/2) through
get_depth_batch, 64 pairs of 1080p per call -- only the batched calls, so that a rocprofv3 --stats of this script is the
kernel mix of that path."""
import sys, os, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import calibrating_amd as ca
from calibrating_amd import synthetic, _native
if len(sys.argv) > 1:  # measurement only: another build of the library (tools/build_dbg.sh)
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
W, H, nb = 1920, 1080, 64
stereo = ca.Stereo.load(synthetic.rig(W, H))
stereo.set_stereo_matching(ca.SemiGlobalBlockMatching({}), max_depth=3.5)
pairs = [synthetic.render_plane_pair(synthetic.rig(W, H), (0.3, 0.1, 1.0), 2.0 + 0.2 * i, seed=i)[:2] for i in range(2)]
B1 = torch.from_numpy(np.stack([pairs[i % 2][0] for i in range(nb)])).cuda()
B2 = torch.from_numpy(np.stack([pairs[i % 2][1] for i in range(nb)])).cuda()
for _ in range(2): stereo.get_depth_batch(B1, B2)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): stereo.get_depth_batch(B1, B2)
torch.cuda.synchronize()
rate = nb * 3 / (time.perf_counter() - t0)
m = stereo.stereo_matching.stereo_sgbm
m.set_profiling(True); stereo.get_depth_batch(B1, B2); torch.cuda.synchronize()
print(json.dumps({"get_depth_batch64_default_plugin_pairs_per_s": rate,
                  "sgbm_stages_ms": {k: round(v, 2) for k, v in m.stage_times_ms().items() if v > 0.01}}))
