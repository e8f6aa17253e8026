#!/usr/bin/env python3
"""Spatial partitioning of the chip between the VALU-bound cost kernel and the HBM-bound aggregation passes
(VERDICT r04 next #3): hipExtStreamCreateWithCUMask streams + CAMD_OPT_PHASES.

    python tools/gpu_cumask.py [--batch 64] [--steps 24] [--mode sgbm|hh] [--lib path]

Variants (all on the bench workload: 1920x1080 RGB, D=128, block 5):
  base2        bench.py's default: two batches in flight on two ordinary streams, every kernel of a batch on its stream
  split N      the same with the two streams masked to N and 256 - N compute units (each batch keeps its partition)
  pipe         cost(k+1) on stream X  ||  aggregation(k) on stream Y, ordinary streams (the pipelining alone)
  pipe N       the same with X masked to N CUs (cost) and Y to the other 256 - N (band passes, WTA, post)
Prints one line per variant and writes gpurun_out/cumask.json.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--mode", default="sgbm")
    ap.add_argument("--lib", default="")
    ap.add_argument("--splits", default="96,128,160,192")
    a = ap.parse_args()
    import torch
    from calibrating_amd import _native
    if a.lib:
        _native.LIB_PATH = os.path.abspath(a.lib)
    import calibrating_amd as ca
    from calibrating_amd import synthetic
    dev = torch.device("cuda", 0)
    W, H, D, cn = 1920, 1080, 128, 3
    P = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=8 * cn * 25, P2=32 * cn * 25, disp12MaxDiff=1,
             preFilterCap=0, uniquenessRatio=10, speckleWindowSize=0, speckleRange=0, mode=1 if a.mode == "hh" else 0)
    nb = a.batch
    L, R = synthetic.rectified_batch_torch(1234, nb, H, W, D, cn, dev)
    lib = _native.lib()

    def masked_stream(lo, hi):
        """CUs [lo, hi) of the driver's enumeration (XCD-major: an equal share of every XCD)."""
        words = (ctypes.c_uint32 * 8)()
        for i in range(lo, hi):
            words[i // 32] |= 1 << (i % 32)
        st = ctypes.c_void_p()
        _native.check(lib.camd_stream_create_cu_mask(words, 8, ctypes.byref(st)), "cu mask")
        return torch.cuda.ExternalStream(st.value, device=dev), st

    ms = [ca.StereoSGBM_create(**P) for _ in range(2)]
    outs = [torch.empty((nb, H, W), dtype=torch.int16, device=dev) for _ in range(2)]
    for m, o in zip(ms, outs):
        m.compute(L, R, out=o)
    torch.cuda.synchronize()
    ref = outs[0].clone()
    results = []

    def report(name, dt, steps, ok):
        r = dict(variant=name, pairs_per_s=nb * steps / dt, ms_per_step=1e3 * dt / steps, same_result=bool(ok))
        results.append(r)
        print("%-12s %8.1f pairs/s  %6.2f ms/step  same=%s" % (name, r["pairs_per_s"], r["ms_per_step"], ok), flush=True)

    def run_two(streams, steps):
        for m in ms:
            m.set_option("phases", 7)
        for k in range(4):
            with torch.cuda.stream(streams[k & 1]):
                ms[k & 1].compute(L, R, out=outs[k & 1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            with torch.cuda.stream(streams[k & 1]):
                ms[k & 1].compute(L, R, out=outs[k & 1])
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def run_pipe(sx, sy, steps):
        ev_cost = [torch.cuda.Event() for _ in range(2)]
        ev_done = [torch.cuda.Event() for _ in range(2)]

        def step(k):
            i = k & 1
            m = ms[i]
            with torch.cuda.stream(sx):
                sx.wait_event(ev_done[i])      # the handle's previous aggregation has finished with C
                m.set_option("phases", 1)
                m.compute(L, R, out=outs[i])
                ev_cost[i].record(sx)
            with torch.cuda.stream(sy):
                sy.wait_event(ev_cost[i])
                m.set_option("phases", 6)
                m.compute(L, R, out=outs[i])
                ev_done[i].record(sy)

        for k in range(4):
            step(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for m in ms:
            m.set_option("phases", 7)
        return dt

    def same():
        return torch.equal(outs[0], ref) and torch.equal(outs[1], ref)

    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    report("base2", run_two([s0, s1], a.steps), a.steps, same())
    report("pipe", run_pipe(s0, s1, a.steps), a.steps, same())
    keep = []
    for n in [int(x) for x in a.splits.split(",") if x]:
        (sa, ha), (sb, hb) = masked_stream(0, n), masked_stream(n, 256)
        keep += [ha, hb]
        report("split %d" % n, run_two([sa, sb], a.steps), a.steps, same())
        report("pipe %d" % n, run_pipe(sa, sb, a.steps), a.steps, same())
    for m in ms:
        m.status()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dict(workload="%dx%d RGB D=%d block 5 mode %s, %d pairs per launch, %d steps" % (W, H, D, a.mode, nb, a.steps),
                   note="split N: two batches in flight, streams masked to N / 256-N CUs; pipe N: cost kernel on N CUs "
                        "|| aggregation + post on 256-N (CAMD_OPT_PHASES), two handles alternating",
                   results=results), open(os.path.join(ROOT, "gpurun_out", "cumask.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
