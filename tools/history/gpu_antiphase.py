"""Experiment: two 64-pair batches on two streams, the second started `offset` ms after the first, so that its VALU-bound
cost kernel runs beside the first batch's band passes (anti-phase) instead of beside its cost kernel (what two free-running
streams settle into: tools/gpu_trace_overlap.sh).  Usage: python tools/gpu_antiphase.py [lib.so]"""
import sys, time, torch
sys.path.insert(0, '.')
if len(sys.argv) > 1:
    import os
    from calibrating_amd import _native
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
import calibrating_amd as ca
from calibrating_amd import synthetic
dev = torch.device('cuda', 0)
P = dict(minDisparity=0, numDisparities=128, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10)
N = 64
L, R = synthetic.rectified_batch_torch(1234, N, 1080, 1920, 128, 3, dev)
ms = [ca.StereoSGBM_create(**P) for _ in range(2)]
outs = [torch.empty((N, 1080, 1920), dtype=torch.int16, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
for i in range(2):
    with torch.cuda.stream(streams[i]):
        ms[i].compute(L, R, out=outs[i])
torch.cuda.synchronize()
t0 = time.perf_counter(); ms[0].compute(L, R, out=outs[0]); torch.cuda.synchronize(); one = time.perf_counter() - t0
print("one batch alone: %.2f ms" % (one * 1e3))
for off in (0, 6, 12, 17, 22, 28, 33):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(streams[0]):
            ms[0].compute(L, R, out=outs[0])
        while time.perf_counter() - t0 < off * 1e-3:  # the host waits; the launches themselves are asynchronous
            pass
        with torch.cuda.stream(streams[1]):
            ms[1].compute(L, R, out=outs[1])
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("second batch %2d ms later: both done after %.2f ms (two alone: %.2f; the later one alone would end at %.2f)" % (off, best * 1e3, 2 * one * 1e3, (off * 1e-3 + one) * 1e3))
