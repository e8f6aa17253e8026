"""Time the Lanczos remap alone (64 x 1080p RGB through a rectification map): python tools/gpu_remap_exp.py [lib.so ...]"""
import sys, os, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 or (len(sys.argv) == 2 and sys.argv[1] == "all"):
    libs = sys.argv[1:] if sys.argv[1] != "all" else [""]
    for l in libs:
        out = subprocess.run([sys.executable, __file__] + ([l] if l else ["-"]), capture_output=True, text=True)
        print(l or "product", out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
    sys.exit(0)
import numpy as np, torch
if len(sys.argv) > 1 and sys.argv[1] != "-":
    from calibrating_amd import _native
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
import calibrating_amd as ca
from calibrating_amd import synthetic, imgproc
W, H, nb = 1920, 1080, 64
dev = torch.device("cuda", 0)
stereo = ca.Stereo.load(synthetic.rig(W, H))
tb = stereo._tables(dev)
pairs = [synthetic.scene_pair(100 + i, W, H, 3) for i in range(4)]
B = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(nb)])).to(dev)
res = {}
for name, b, m in (("rgb_map1", B, "map1"), ("rgb_map2", B, "map2"), ("gray_map1", B[..., 1:2].contiguous(), "map1")):
    for _ in range(2): imgproc.remap(b, tb[m + "x"], tb[m + "y"], imgproc.INTER_LANCZOS4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): imgproc.remap(b, tb[m + "x"], tb[m + "y"], imgproc.INTER_LANCZOS4)
    e1.record(); torch.cuda.synchronize()
    res[name + "_ms_per_64"] = round(e0.elapsed_time(e1) / 5, 3)
for m in ("map1", "map2"):
    mx, my = tb[m + "x"], tb[m + "y"]
    res[m + "_range"] = [float(mx.min()), float(mx.max()), float(my.min()), float(my.max())]
print(json.dumps(res))
