# experiment: does running two half-batches on two HIP streams overlap the VALU-bound cost kernel with the
# HBM-bound aggregation passes?
import sys, time, torch
sys.path.insert(0, '.')
import calibrating_amd as ca
from calibrating_amd import synthetic
dev = torch.device('cuda', 0)
P = dict(minDisparity=0, numDisparities=128, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10)
N = 64
L, R = synthetic.rectified_batch_torch(1234, N, 1080, 1920, 128, 3, dev)
def run(nstreams, reps=3, per=None):
    per = per or N // nstreams
    ms = [ca.StereoSGBM_create(**P) for _ in range(nstreams)]
    outs = [torch.empty((per, 1080, 1920), dtype=torch.int16, device=dev) for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    def step():
        for i in range(nstreams):
            with torch.cuda.stream(streams[i]):
                lo = (i * per) % N
                ms[i].compute(L[lo:lo+per], R[lo:lo+per], out=outs[i])
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print('streams', nstreams, 'pairs per stream', per, 'pairs/s %.1f' % (nstreams * per / dt), 'ms/step %.2f' % (dt * 1e3))
    del ms, outs
    torch.cuda.empty_cache()
for n, per in ((1, 64), (2, 32), (2, 64), (3, 64), (4, 32)):
    run(n, per=per)
