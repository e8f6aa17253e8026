#!/usr/bin/env python3
"""Round 6: which kernels of consecutive batches should run side by side?  (CAMD_OPT_PHASES: 1 = cost volume, 2 = first
aggregation pass, 4 = last pass + post.)  1920x1080 RGB, D=128, block 5, 64 pairs per launch, two handles alternating.

  serial       one stream, one batch after the other
  free2        bench.py's default: two batches in flight on two streams, no ordering between them
  cost|last    k_cost of batch k+1 starts when the FIRST pass of batch k has finished: the VALU-bound cost kernel beside
               the HBM-bound row-parallel last pass; the first pass (which has little slack of either kind) runs alone
  cost|first   k_cost of batch k+1 starts with the first pass of batch k; the last pass runs alone
  cost|both    k_cost of batch k+1 free-running beside both passes of batch k (round 5's 'pipe')
each with the aggregation stream at normal / high priority (hipStreamCreateWithPriority).
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--mode", default="sgbm")
    ap.add_argument("--out", default="r06_pipeline.json")
    ap.add_argument("--only-resident", action="store_true", help="run only the cost|last variant of each --resident value (for traces)")
    ap.add_argument("--resident", default="", help="comma list of hex values 0xAB: A cost / B last-pass workgroups per CU")
    a = ap.parse_args()
    import torch
    import calibrating_amd as ca
    from calibrating_amd import synthetic
    dev = torch.device("cuda", 0)
    W, H, D, cn = 1920, 1080, 128, 3
    P = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=8 * cn * 25, P2=32 * cn * 25, disp12MaxDiff=1,
             preFilterCap=0, uniquenessRatio=10, speckleWindowSize=0, speckleRange=0, mode=1 if a.mode == "hh" else 0)
    nb = a.batch
    L, R = synthetic.rectified_batch_torch(1234, nb, H, W, D, cn, dev)
    ms = [ca.StereoSGBM_create(**P) for _ in range(2)]
    outs = [torch.empty((nb, H, W), dtype=torch.int16, device=dev) for _ in range(2)]
    for m, o in zip(ms, outs):
        m.set_option("path", 2)
        m.compute(L, R, out=o)
    torch.cuda.synchronize()
    ref = outs[0].clone()
    results = []

    def report(name, dt, steps):
        ok = torch.equal(outs[0], ref) and torch.equal(outs[1], ref)
        r = dict(variant=name, pairs_per_s=nb * steps / dt, ms_per_step=1e3 * dt / steps, same_result=bool(ok))
        results.append(r)
        print("%-26s %8.1f pairs/s  %6.2f ms/step  same=%s" % (name, r["pairs_per_s"], r["ms_per_step"], ok), flush=True)
        for o in outs:
            o.zero_()

    def timed(step, steps):
        for k in range(4):
            step(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def run_free(streams, steps):
        for m in ms:
            m.set_option("phases", 7)

        def step(k):
            with torch.cuda.stream(streams[k % len(streams)]):
                ms[k & 1].compute(L, R, out=outs[k & 1])
        return timed(step, steps)

    def run_pipe(sc, sa, gate, steps):
        """cost on stream sc, aggregation on sa; gate: 'first_done' / 'first_start' / None = when may cost(k+1) start
        relative to batch k's passes."""
        ev_cost = [torch.cuda.Event() for _ in range(2)]
        ev_done = [torch.cuda.Event() for _ in range(2)]   # the handle's last pass has finished with C
        ev_gate = [torch.cuda.Event()]                     # what the NEXT cost kernel waits for
        ev_gate[0].record(sa)

        def step(k):
            i = k & 1
            m = ms[i]
            with torch.cuda.stream(sc):
                sc.wait_event(ev_done[i])
                if gate:
                    sc.wait_event(ev_gate[0])
                m.set_option("phases", 1)
                m.compute(L, R, out=outs[i])
                ev_cost[i].record(sc)
            with torch.cuda.stream(sa):
                sa.wait_event(ev_cost[i])
                if gate == "first_start":
                    ev_gate[0] = torch.cuda.Event(); ev_gate[0].record(sa)
                m.set_option("phases", 2)
                m.compute(L, R, out=outs[i])
                if gate == "first_done":
                    ev_gate[0] = torch.cuda.Event(); ev_gate[0].record(sa)
                m.set_option("phases", 4)
                m.compute(L, R, out=outs[i])
                ev_done[i].record(sa)
        dt = timed(step, steps)
        for m in ms:
            m.set_option("phases", 7)
        return dt

    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    hi = torch.cuda.Stream(priority=-1)
    if not a.only_resident:
        report("serial", run_free([s0], a.steps), a.steps)
        report("free2", run_free([s0, s1], a.steps), a.steps)
    if not a.resident:
        for gate, name in (("first_done", "cost|last"), ("first_start", "cost|first"), (None, "cost|both")):
            report(name, run_pipe(s0, s1, gate, a.steps), a.steps)
            report(name + " agg-prio", run_pipe(s0, hi, gate, a.steps), a.steps)
            report(name + " cost-prio", run_pipe(hi, s1, gate, a.steps), a.steps)
    # CAMD_OPT_RESIDENT: a cost / b last-pass persistent workgroups per CU, so that both launches stay resident together
    for r in [int(x, 16) for x in a.resident.split(",") if x]:
        for m in ms:
            m.set_option("resident", r)
        tag = "resident %d:%d " % (r >> 4, r & 15)
        if not a.only_resident:
            report(tag + "serial", run_free([s0], a.steps), a.steps)
        report(tag + "cost|last", run_pipe(s0, s1, "first_done", a.steps), a.steps)
        if not a.only_resident:
            report(tag + "cost|last agg-prio", run_pipe(s0, hi, "first_done", a.steps), a.steps)
        for m in ms:
            m.set_option("resident", 0)
    for m in ms:
        m.status()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dict(workload="%dx%d RGB D=%d block 5 mode %s, %d pairs per launch, %d steps" % (W, H, D, a.mode, nb, a.steps),
                   results=results), open(os.path.join(ROOT, "gpurun_out", a.out), "w"), indent=1)


if __name__ == "__main__":
    main()
