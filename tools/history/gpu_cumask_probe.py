#!/usr/bin/env python3
"""Which compute units do the bits of a hipExtStreamCreateWithCUMask mask name?  Times the cost kernel (VALU-bound) and the
aggregation passes (band passes: latency / HBM-bound) of one 64-pair batch on streams masked with different bit patterns
of equal population.  -> gpurun_out/r05_cumask_probe.json"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import calibrating_amd as ca
from calibrating_amd import _native, synthetic
dev = torch.device("cuda", 0)
W, H, D, cn, nb = 1920, 1080, 128, 3, 64
P = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10)
L, R = synthetic.rectified_batch_torch(1234, nb, H, W, D, cn, dev)
m = ca.StereoSGBM_create(**P)
out = torch.empty((nb, H, W), dtype=torch.int16, device=dev)
m.compute(L, R, out=out); torch.cuda.synchronize()
lib = _native.lib()
patterns = {
    "all 256": lambda i: True,
    "bits 0..127": lambda i: i < 128,
    "even bits": lambda i: i % 2 == 0,
    "i % 8 < 4": lambda i: i % 8 < 4,
    "(i // 32) % 2 == 0": lambda i: (i // 32) % 2 == 0,
    "(i // 16) % 2 == 0": lambda i: (i // 16) % 2 == 0,
    "bits 0..111": lambda i: i < 112,
    "i % 16 < 7 (112)": lambda i: i % 16 < 7,
    "bits 0..63": lambda i: i < 64,
    "i % 4 == 0 (64)": lambda i: i % 4 == 0,
    "i % 8 < 2 (64)": lambda i: i % 8 < 2,
}
res = {}
for name, f in patterns.items():
    words = (ctypes.c_uint32 * 8)()
    n = 0
    for i in range(256):
        if f(i):
            words[i // 32] |= 1 << (i % 32); n += 1
    st = ctypes.c_void_p()
    _native.check(lib.camd_stream_create_cu_mask(words, 8, ctypes.byref(st)))
    s = torch.cuda.ExternalStream(st.value, device=dev)
    row = {"cus": n}
    for label, ph in (("cost_ms", 1), ("aggregation_ms", 6)):
        m.set_option("phases", ph)
        with torch.cuda.stream(s):
            m.compute(L, R, out=out)
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                m.compute(L, R, out=out)
            s.synchronize()
        row[label] = (time.perf_counter() - t0) / 3 * 1e3
    m.set_option("phases", 7)
    res[name] = row
    print("%-22s %3d CUs  cost %7.2f ms  aggregation %7.2f ms" % (name, n, row["cost_ms"], row["aggregation_ms"]), flush=True)
    torch.cuda.synchronize()
    lib.camd_stream_destroy(st)
m.status()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r05_cumask_probe.json"), "w"), indent=1)
