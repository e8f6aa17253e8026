# experiment: staggered two-stream schedule of two half-batches inside one "step" (what an internal split of
# camd_sgbm_compute could achieve): stream B starts when stream A's cost kernel is done (approximated by a delay event)
import sys, time, torch
sys.path.insert(0, '.')
import calibrating_amd as ca
from calibrating_amd import synthetic
dev = torch.device('cuda', 0)
P = dict(minDisparity=0, numDisparities=128, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L, R = synthetic.rectified_batch_torch(1234, N, 1080, 1920, 128, 3, dev)

def bench(name, step, reps=5):
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
        torch.cuda.synchronize()   # one step = one call that must be complete (no overlap across steps)
    dt = (time.perf_counter() - t0) / reps
    print('%-40s pairs/s %.1f  ms/step %.2f' % (name, N / dt, dt * 1e3))

m64 = ca.StereoSGBM_create(**P); o64 = torch.empty((N, 1080, 1920), dtype=torch.int16, device=dev)
bench('one stream, 64 pairs', lambda: m64.compute(L, R, out=o64))
del m64

for parts in (2, 4):
    per = N // parts
    ms = [ca.StereoSGBM_create(**P) for _ in range(parts)]
    outs = [torch.empty((per, 1080, 1920), dtype=torch.int16, device=dev) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    def simultaneous():
        for i in range(parts):
            with torch.cuda.stream(streams[i]):
                ms[i].compute(L[i*per:(i+1)*per], R[i*per:(i+1)*per], out=outs[i])
    bench('%d streams x %d pairs, simultaneous' % (parts, per), simultaneous)
    # stagger: part i+1 may start only when part i has finished a small "marker" kernel queued right after its compute
    # call's first kernels ... we cannot insert an event inside compute(), so emulate: split compute by running part
    # i+1's stream behind a sleep-free dependency on a dummy op enqueued BEFORE part i's compute (no stagger) vs
    # AFTER a fraction of work: use two handles per part? -> simplest emulation: chain starts with torch.cuda._sleep
    for frac_ms in (4, 8, 12):
        cycles = int(frac_ms * 1e-3 * 2.1e9)
        def staggered():
            for i in range(parts):
                with torch.cuda.stream(streams[i]):
                    if i: torch.cuda._sleep(cycles * i)
                    ms[i].compute(L[i*per:(i+1)*per], R[i*per:(i+1)*per], out=outs[i])
        bench('%d streams x %d pairs, start offset %d ms' % (parts, per, frac_ms), staggered)
    del ms, outs
