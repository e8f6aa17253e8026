"""The drop-in call as the reference's users make it: Stereo.get_depth(ndarray, ndarray) -> dict of ndarrays, one pair
per call.  Wall time per call at 1080p and VGA, and where it goes."""
import sys, os, time, json, cProfile, pstats, io
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import calibrating_amd as ca
from calibrating_amd import synthetic, hostio
res = {}
if "--bind" in sys.argv:
    res["bound_to_cpus"] = hostio.bind_near_gpu(0)
for tag, W, H, D in (("1080p_d128", 1920, 1080, 128), ("vga_d64", 640, 480, 64)):
    stereo = ca.Stereo.load(synthetic.rig(W, H))
    cfg = dict(max_size=max(W, H), minDisparity=0, numDisparities=D, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
               uniquenessRatio=10, speckleWindowSize=100, speckleRange=2)
    stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(cfg), max_depth=3.5)
    i1, i2 = synthetic.scene_pair(9, W, H, 3)
    for _ in range(3): out = stereo.get_depth(i1, i2)
    t0 = time.perf_counter()
    for _ in range(20): out = stereo.get_depth(i1, i2)
    res[tag + "_numpy_in_out_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    res[tag + "_output_MB"] = sum(v.nbytes for v in out.values() if isinstance(v, np.ndarray)) / 1e6
    for name, keys in (("unrectify_depth_only", ("unrectify_depth",)), ("depths_only", ("rectify_depth", "unrectify_depth"))):
        for _ in range(3): out = stereo.get_depth(i1, i2, keys=keys)
        t0 = time.perf_counter()
        for _ in range(20): out = stereo.get_depth(i1, i2, keys=keys)
        res[tag + "_numpy_%s_ms" % name] = (time.perf_counter() - t0) / 20 * 1e3
        res[tag + "_%s_MB" % name] = sum(v.nbytes for v in out.values()) / 1e6
    # one pair per call, but up to `depth` calls in flight (Stereo.get_depth_async): calls per second
    import collections
    for depth in (2, 3):
        for name, keys in (("full_dict", None), ("unrectify_depth_only", ("unrectify_depth",))):
            q = collections.deque()
            n = 40
            for k in range(n + 6):
                if k == 6:
                    while q: q.popleft().result()
                    t0 = time.perf_counter()
                q.append(stereo.get_depth_async(i1, i2, keys=keys))
                if len(q) >= depth: out = q.popleft().result()
            while q: out = q.popleft().result()
            res[tag + "_async_%d_in_flight_%s_calls_per_s" % (depth, name)] = n / (time.perf_counter() - t0)
    t1, t2 = torch.from_numpy(i1).cuda(), torch.from_numpy(i2).cuda()
    for _ in range(3): stereo.get_depth(t1, t2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): stereo.get_depth(t1, t2)
    torch.cuda.synchronize(); res[tag + "_device_resident_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    if tag.startswith("1080p") and "--profile" in sys.argv:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(10): stereo.get_depth(i1, i2)
        pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12); print(s.getvalue()[:3500])
# one matcher alternating between two image sizes (StereoSGBM keeps a handle per shape since round 5)
p = dict(minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10)
pa = [torch.from_numpy(x).cuda() for x in synthetic.rectified_pair(1, 480, 640, 64, 3)]
pb = [torch.from_numpy(x).cuda() for x in synthetic.rectified_pair(2, 600, 800, 64, 3)]
for cache in (4, 1):
    m = ca.StereoSGBM_create(**p)
    m.HANDLE_CACHE = cache
    m.compute(*pa); m.compute(*pb); torch.cuda.synchronize()
    n0 = ca.StereoSGBM.creates; t0 = time.perf_counter()
    for k in range(10): m.compute(*(pa, pb)[k & 1])
    torch.cuda.synchronize()
    res["alternating_640x480_800x600_ms_per_call_cache%d" % cache] = (time.perf_counter() - t0) / 10 * 1e3
    res["alternating_creates_per_10_calls_cache%d" % cache] = ca.StereoSGBM.creates - n0
print(json.dumps({k: round(v, 3) if isinstance(v, float) else v for k, v in res.items()}, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/numpy_latency.json", "w"), indent=1)
