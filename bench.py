#!/usr/bin/env python3
"""bench.py -- stereo pairs/s at 1920x1080, numDisparities=128 on MI355X (BASELINE.json metric).

A step = one pass of the SGBM hot path (BT cost volume -> path aggregation -> WTA / uniqueness /
LR check -> median) over one batch of synthetic rectified pairs that are already resident in HBM.
Consecutive steps alternate over --in-flight (default 2) independent sets of {inputs, handle, output, HIP stream}, the
way a streaming deployment double-buffers: the VALU-bound cost kernel of one batch then overlaps the HBM-bound last
pass of the previous one (+8 % over one batch at a time, which `also.single_stream_pairs_per_s` reports).  The
timed region is exactly --steps steps between barrier + synchronize.
Independent pairs shard across ranks with no data-path collective ("scaling": "weak": every rank
processes its own batch of --batch pairs); the only RCCL traffic is the one-time broadcast of the rig's
remap tables and the end-of-run reduction of timings / checksums.

    python bench.py [--gpus N --steps K --warmup W]

With --gpus N > 1 and no RANK in the environment the script launches itself under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, RCCL); it exits non-zero with
a message when fewer than N GPUs are visible.  Under an external torchrun it reads RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment.

Default workload = BASELINE.json configs[1]/[2]: 1920x1080 RGB pairs (the reference feeds RGB), D=128,
blockSize=5, cv2's default MODE_SGBM (5 paths: the mode the reference's cv2.StereoSGBM_create call
selects), 64 pairs per GPU per step (512 pairs / 8 GPUs).  --mode hh runs the 8-path MODE_HH; its
throughput is also reported in "also" on every default run.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
PMC_PROFILE = "profiles/r02_pmc_traffic%s.json"  # committed rocprofv3 --pmc summary the `traffic` fields come from


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80, help="timed steps (default: >= 3 s of GPU work)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU per step")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="batches in flight per GPU: consecutive steps alternate over this many handle + stream sets, so "
                         "the VALU-bound cost kernel of one batch overlaps the HBM-bound last pass of the previous one")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--disparities", type=int, default=128)
    ap.add_argument("--block", type=int, default=5)
    ap.add_argument("--channels", type=int, default=3, help="3 = RGB as the reference feeds SGBM, 1 = gray")
    ap.add_argument("--mode", default="sgbm", choices=["sgbm", "hh"],
                    help="sgbm = 5-path MODE_SGBM (the reference's call), hh = 8-path MODE_HH")
    ap.add_argument("--path", type=int, default=0, help="aggregation path: 0 auto, 1 scans, 2 band passes, 3 concurrent")
    ap.add_argument("--cost", type=int, default=0, help="cost-volume kernels: 0 auto (fused), 1 fused k_cost, 2 k_hsum + k_vsum")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary measurements (other mode, gray, PCIe)")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="strips in the CPU baseline sample (0 = two per thread)")
    ap.add_argument("--lib", default="", help="measurement only: load this build of libcalibrating_amd.so (A/B kernels)")
    return ap.parse_args(argv)


def sgbm_params(a, mode=None, channels=None):
    cn, bs = (a.channels if channels is None else channels), a.block
    mode = a.mode if mode is None else mode
    return dict(minDisparity=0, numDisparities=a.disparities, blockSize=bs, P1=8 * cn * bs * bs,
                P2=32 * cn * bs * bs, disp12MaxDiff=1, preFilterCap=0, uniquenessRatio=10,
                speckleWindowSize=0, speckleRange=0, mode=1 if mode == "hh" else 0)


def algorithmic_bytes_per_pair(W, H, D, cn, minD=0):
    """B_alg = 2*V + IO, V = H*W1*D*2, IO = 2*H*W*cn + 2*H*W."""
    maxD = minD + D
    W1 = (W + min(minD, 0)) - max(maxD, 0)
    V = H * W1 * D * 2
    return 2 * V + 2 * H * W * cn + 2 * H * W, V


# stage (hipEvent bracket inside the library) -> (kernel it times, algorithmic HBM bytes per pair as f(V, HW, cn),
# substrings that identify the kernel in a rocprofv3 summary)
def stage_table(V, HW, cn, mode):
    last = "k_band last pass (%s + WTA: C and S in)" % ("4 directions" if mode == "hh" else "1 direction")
    return {
        "cost": ("k_cost (BT cost + KxK box sum -> C, fused)", V + 2 * HW * cn, ("k_cost",)),
        "hsum": ("k_hsum (BT cost + horizontal box sum)", V + 2 * HW * cn, ("k_hsum",)),
        "vsum": ("k_vsum (vertical box sum + P2 -> C)", 2 * V, ("k_vsum",)),
        "scan": ("k_band first pass (4 directions: C in, S out)", 2 * V, ("k_band", ", true, 0,")),
        "scan_last": (last, 2 * V, ("k_band", ", 2,")),
        "wta": ("k_lrcheck", 12 * HW, ("k_lrcheck",)),
        "median_speckle": ("k_median3", 4 * HW, ("k_median3",)),
    }


def cpu_baseline(a, params):
    """The CPU oracle (scalar C port of cv2.StereoSGBM, oracle/sgbm_ref.c) on ALL of this host's cores."""
    import oracle
    from calibrating_amd import synthetic
    oracle.build()
    ncpu = os.cpu_count() or 1
    threads = ncpu
    # bounded sample (~10-30 s of CPU work): full-width strips of a quarter of the rows (SGBM cost is linear in
    # rows), one strip per thread; scaled back to whole pairs below
    hs = max(a.height // 4, 16)
    if a.mode == "hh":  # the two-pass mode keeps two whole strip volumes per thread: bound the host memory (32 GB)
        vol = 2 * 2 * hs * max(a.width - a.disparities, 1) * a.disparities
        threads = max(1, min(threads, int(32e9 // vol)))
    n = a.cpu_pairs or threads
    base_l, base_r = synthetic.rectified_pair(seed=1234, H=hs, W=a.width, D=a.disparities, cn=a.channels)
    L = np.stack([np.roll(base_l, 17 * i, axis=0) for i in range(n)])  # distinct strips: vertical rolls
    R = np.stack([np.roll(base_r, 17 * i, axis=0) for i in range(n)])
    # this leg runs on ALL host CPUs: lift the binding to the GPU's NUMA node (hostio.bind_near_gpu) while it lasts --
    # OpenMP threads inherit the affinity of the thread that starts them
    bound = os.sched_getaffinity(0)
    try:
        os.sched_setaffinity(0, range(ncpu))
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0))
    t0 = time.perf_counter()
    oracle.sgbm_compute_batch(L, R, nthreads=threads, **params)
    dt = time.perf_counter() - t0
    # the port streams a volume per thread and is bound by host memory bandwidth long before it runs out of cores:
    # a second, short sample on an eighth of the threads is reported beside the all-core figure
    fewer = {}
    if threads >= 16 and not a.cpu_pairs:
        t8 = threads // 8
        t1 = time.perf_counter()
        oracle.sgbm_compute_batch(L[:t8], R[:t8], nthreads=t8, **params)
        fewer = {str(t8): (t8 * hs / a.height) / (time.perf_counter() - t1)}
    os.sched_setaffinity(0, bound)
    pairs = n * hs / a.height
    return dict(value=pairs / dt, unit="pairs/s", cores=threads, host_cpu_count=ncpu, kind="port",
                pairs_per_s_at_fewer_threads=fewer,
                sample="%d strips of %dx%d (= %.2f pairs of %dx%d) D=%d cn=%d mode=%s, scalar C port "
                       "oracle/sgbm_ref.c, %d OpenMP threads across strips on %d usable CPUs (os.cpu_count() = %d), "
                       "%.1f s; cv2 itself is not installed on this box"
                       % (n, a.width, hs, pairs, a.width, a.height, a.disparities, a.channels, a.mode,
                          threads, usable, ncpu, dt))


def gpu_steps(matcher, left, right, out, steps, warmup, distributed=False):
    """(seconds, {stage: ms summed over the timed steps}) -- parallel_pairs.timed_steps around compute()."""
    import torch
    from calibrating_amd.parallel_pairs import timed_steps
    stage_ms = {}

    def step():
        matcher.compute(left, right, out=out)

    def timed_step():
        matcher.compute(left, right, out=out)
        for k, v in matcher.stage_times_ms().items():  # hipEvents on the compute stream, read after the fact
            stage_ms[k] = stage_ms.get(k, 0.0) + v

    timed_steps(step, 0, warmup, torch.cuda.synchronize, False)
    dt = timed_steps(timed_step, steps, 0, torch.cuda.synchronize, distributed)
    return dt, stage_ms


def pipelined_steps(matchers, streams, lefts, rights, outs, steps, warmup, distributed=False):
    """seconds for `steps` steps, step k on handle / stream k % len(matchers); nothing synchronises between steps."""
    import torch
    from calibrating_amd.parallel_pairs import timed_steps
    n = len(matchers)
    k = [0]
    # set-up, not a step: the first compute on a handle touches its freshly allocated workspace (seconds for 80 GB),
    # so every set runs once before the W warm-up steps -- whatever W is
    for i in range(n):
        with torch.cuda.stream(streams[i]):
            matchers[i].compute(lefts[i], rights[i], out=outs[i])
    torch.cuda.synchronize()

    def step():
        i = k[0] % n
        k[0] += 1
        with torch.cuda.stream(streams[i]):
            matchers[i].compute(lefts[i], rights[i], out=outs[i])

    return timed_steps(step, steps, warmup, torch.cuda.synchronize, distributed)


def pcie_inclusive(matcher, left, right, out, steps, streams=()):
    """pairs/s when every step's inputs come from (pinned) host memory and its disparities go back to it:
    H2D of step k+1 and D2H of step k-1 overlap the compute of step k on three streams, two device buffers."""
    import torch
    nb = left.shape[0]
    hl, hr = left.cpu().pin_memory(), right.cpu().pin_memory()
    ho = torch.empty(out.shape, dtype=out.dtype).pin_memory()
    dl = [left, torch.empty_like(left)]
    dr = [right, torch.empty_like(right)]
    do = [out, torch.empty_like(out)]
    # the copy streams REUSE the side streams of the timed region where they exist: HIP maps streams onto a handful of
    # hardware queues, and a fifth live stream would share one with (and serialise against) the compute stream
    pool = list(streams) + [torch.cuda.Stream() for _ in range(max(0, 2 - len(streams)))]
    s_in, s_out, s_cmp = pool[0], pool[1], torch.cuda.current_stream()
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_cmp = [torch.cuda.Event() for _ in range(2)]
    ev_out = [torch.cuda.Event() for _ in range(2)]

    def run(k):
        b = k & 1
        with torch.cuda.stream(s_in):
            s_in.wait_event(ev_cmp[b])        # buffer b's previous compute has consumed its inputs
            dl[b].copy_(hl, non_blocking=True)
            dr[b].copy_(hr, non_blocking=True)
            ev_in[b].record(s_in)
        s_cmp.wait_event(ev_in[b])
        s_cmp.wait_event(ev_out[b])           # buffer b's previous result has left
        matcher.compute(dl[b], dr[b], out=do[b])
        ev_cmp[b].record(s_cmp)
        with torch.cuda.stream(s_out):
            s_out.wait_event(ev_cmp[b])
            ho.copy_(do[b], non_blocking=True)
            ev_out[b].record(s_out)

    for k in range(2):
        run(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        run(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    bytes_per_pair = (hl[0].numel() + hr[0].numel()) + ho[0].numel() * 2
    return dict(pairs_per_s=nb * steps / dt, host_bytes_per_pair=bytes_per_pair,
                host_link_GBs=bytes_per_pair * nb * steps / dt / 1e9,
                note="pinned host buffers, H2D + compute + D2H overlapped on 3 streams, %d steps of %d pairs" % (steps, nb))


def self_launch(a):
    """--gpus N > 1 without a launcher: run N ranks of this script under torch.distributed.run."""
    import torch
    have = torch.cuda.device_count()
    if have < a.gpus:
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) visible; refusing to report a "
                         "multi-GPU number from fewer devices\n" % (a.gpus, have))
        sys.exit(2)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        self_launch(a)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" in os.environ and a.gpus not in (1, world):
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    if local_rank >= torch.cuda.device_count():
        sys.exit("bench.py: rank %d has no GPU (LOCAL_RANK %d, %d visible)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from calibrating_amd import hostio
    cpus_near_gpu = hostio.bind_near_gpu(local_rank)
    # CAMD_BENCH_FORCE_DIST=1 exercises the RCCL path (init, table broadcast, barrier, reductions) with a
    # single rank, e.g. under `python -m torch.distributed.run --nproc-per-node 1`
    distributed = world > 1 or (os.environ.get("CAMD_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ["NCCL_DEBUG"] = os.environ.get("CAMD_NCCL_DEBUG", "WARN")  # keep RCCL's banner off stdout
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if a.lib:
        from calibrating_amd import _native
        _native.LIB_PATH = os.path.abspath(a.lib)
    import calibrating_amd as ca
    from calibrating_amd import synthetic
    from calibrating_amd.parallel_pairs import aggregate, broadcast_tables, ranks_agree, shard_range

    params = sgbm_params(a)
    # this rank's shard of the global pair list: pairs [lo, hi) of world*batch
    lo, hi = shard_range(world * a.batch, world, rank)
    nb = hi - lo
    nfl = max(1, min(a.in_flight, a.steps))
    lefts, rights, outs, matchers, streams = [], [], [], [], []
    for i in range(nfl):  # every batch in flight has its own inputs, handle (workspace), output and stream
        l, r = synthetic.rectified_batch_torch(1234 + rank + 1000 * i, nb, a.height, a.width, a.disparities,
                                               a.channels, dev)
        m = ca.StereoSGBM_create(**params)
        m.set_option("path", a.path)
        m.set_option("cost", a.cost)
        lefts.append(l); rights.append(r); matchers.append(m)
        outs.append(torch.empty((nb, a.height, a.width), dtype=torch.int16, device=dev))
        streams.append(torch.cuda.Stream(device=dev))
    left, right, out, matcher = lefts[0], rights[0], outs[0], matchers[0]

    # (1) kernel characterisation: a few steps on ONE stream with the library's hipEvents around every kernel
    matcher.set_profiling(True)
    kprof = max(1, min(a.steps, 5))
    dt1, stage_ms = gpu_steps(matcher, left, right, out, kprof, a.warmup)
    matcher.set_profiling(False)
    prof_steps = kprof
    # (2) the timed region: exactly --steps steps, `nfl` batches in flight, barrier + synchronize on both sides
    dt = pipelined_steps(matchers, streams, lefts, rights, outs, a.steps, a.warmup, distributed)
    for m in matchers:
        m.status()  # raises if a device-side bounded wait timed out
    checksum = sum(int(o.to(torch.int64).sum().item()) for o in outs)
    agg = aggregate(nb * a.steps, dt, checksum, dev, distributed)
    value = agg["total_pairs"] / agg["seconds"]
    single_stream = nb * kprof / dt1
    del matchers[1:], lefts[1:], rights[1:], outs[1:], m, l, r  # (the loop variables hold the last set alive)
    torch.cuda.empty_cache()  # (the host-link rate measured below drops when much more HBM is allocated)

    rccl = None
    if distributed:
        # one-time table broadcast (rank 0 owns the rig), installed into every rank's Stereo; then every rank runs
        # the full get_depth on the SAME two pairs through the broadcast tables and the ranks compare checksums
        t0 = time.perf_counter()
        rig_dict = synthetic.rig(a.width, a.height)
        bundle = ca.Stereo.load(rig_dict).table_bundle() if rank == 0 else None
        tables = broadcast_tables(bundle, dev, src=0)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
        stereo = ca.Stereo.load(rig_dict).install_tables(tables, dev)
        stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(dict(params, max_size=max(a.width, a.height))),
                                   max_depth=20.0)
        imgs = torch.stack([torch.from_numpy(np.ascontiguousarray(im)) for im in
                            synthetic.scene_pair(77, a.width, a.height, 3)]).to(dev)
        res = stereo.get_depth_batch(imgs[:1].repeat(2, 1, 1, 1), imgs[1:].repeat(2, 1, 1, 1))
        dsum = int(torch.nan_to_num(res["unrectify_depth"]).mul(1e4).round().to(torch.int64).sum().item())
        rccl = dict(backend="nccl (RCCL)", world_size=world, table_bytes=int(sum(t.numel() * t.element_size()
                                                                                  for t in tables.values())),
                    broadcast_s=t_bcast, get_depth_checksum=dsum, ranks_agree=bool(ranks_agree(dsum, dev)),
                    per_rank=[dict(pairs=p, seconds=s, disparity_checksum=c) for p, s, c in agg["per_rank"]])
        del stereo, tables, res, imgs

    also = {}
    if rank == 0 and world == 1 and not a.no_also:
        k2 = max(3, min(a.steps, 10))
        also["note"] = "every figure in `also` is measured with ONE batch in flight on one stream"
        also["single_stream_pairs_per_s"] = single_stream
        if os.environ.get("CAMD_BENCH_DEBUG"):
            free, total = torch.cuda.mem_get_info()
            sys.stderr.write("before pcie_inclusive: %.1f GB of %.1f GB free\n" % (free / 1e9, total / 1e9))
        also["pcie_inclusive"] = pcie_inclusive(matcher, left, right, out, k2, streams)
        del matcher
        matchers.clear()
        # the other aggregation mode on the same inputs, and the gray variant of the headline mode
        other = "hh" if a.mode == "sgbm" else "sgbm"
        m2 = ca.StereoSGBM_create(**sgbm_params(a, other))
        m2.set_profiling(True)
        m2.set_option("path", a.path)
        m2.set_option("cost", a.cost)
        d2, _ = gpu_steps(m2, left, right, out, k2, 1)
        also["%s_%s_pairs_per_s" % ("rgb" if a.channels == 3 else "gray", other)] = nb * k2 / d2
        del m2
        if a.channels == 3:
            m3 = ca.StereoSGBM_create(**sgbm_params(a, channels=1))
            m3.set_profiling(True)
            m3.set_option("path", a.path)
            m3.set_option("cost", a.cost)
            gl, gr = left[..., 1].contiguous(), right[..., 1].contiguous()
            d3, _ = gpu_steps(m3, gl, gr, out, k2, 1)
            also["gray_%s_pairs_per_s" % a.mode] = nb * k2 / d3
            del m3, gl, gr
            # the whole Stereo.get_depth path around the same matcher (rectify x2 -> SGBM -> disp_to_depth -> unrectify
            # -> undistort) on a synthetic rig of the same size, one batch per call
            torch.cuda.empty_cache()
            stereo = ca.Stereo.load(synthetic.rig(a.width, a.height))
            stereo.set_stereo_matching(ca.SemiGlobalBlockMatching(dict(params, max_size=max(a.width, a.height))),
                                       max_depth=20.0)
            stereo.get_depth_batch(left, right)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                stereo.get_depth_batch(left, right)
            torch.cuda.synchronize()
            also["get_depth_batch_pairs_per_s"] = nb * 3 / (time.perf_counter() - t0)
            del stereo

    if rank == 0:
        b_alg, V = algorithmic_bytes_per_pair(a.width, a.height, a.disparities, a.channels)
        table = stage_table(V, a.width * a.height, a.channels, a.mode)
        kernel_ms_step = sum(stage_ms.values()) / prof_steps   # one batch alone on the GPU
        gpu_ms_step = agg["seconds"] / a.steps * 1e3             # the timed region (batches in flight overlap)
        # HBM traffic per kernel from the committed PMC profile of the same kernels (separate rocprofv3 --pmc
        # passes, 2*FETCH_SIZE + WRITE_SIZE per the gfx950 correction); null when no profile matches this workload
        pmc_path = os.path.join(ROOT, PMC_PROFILE % ("_hh" if a.mode == "hh" else ""))
        pmc = None
        if (a.channels == 3 and (a.width, a.height, a.disparities, a.block) == (1920, 1080, 128, 5)
                and os.path.exists(pmc_path)):
            pmc = json.load(open(pmc_path))

        def pmc_bytes_per_pair(keys):
            if not pmc:
                return None
            hit = [v["hbm_bytes_per_pair"] for k, v in pmc["kernels"].items() if all(s in k for s in keys)]
            return hit[0] if len(hit) == 1 else None

        kernels = {}
        for st, ms_sum in stage_ms.items():
            ms = ms_sum / prof_steps
            if st not in table or ms < 0.02:  # (an empty stage bracket still measures a few microseconds)
                continue
            name, per_pair, keys = table[st]
            tr = pmc_bytes_per_pair(keys)
            ach = per_pair * nb / (ms * 1e-3) / 1e9
            kernels[st] = {"kernel": name, "avg_ms_per_launch": ms, "algorithmic_bytes_per_launch": float(per_pair * nb),
                           "achieved": ach, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": (tr * nb) if tr is not None else None}
        dom = max(kernels, key=lambda k: kernels[k]["avg_ms_per_launch"]) if kernels else None
        # whole-step traffic = every kernel of the committed PMC profile (incl. the small init / check kernels)
        traffic_total = sum(v["hbm_bytes_per_pair"] for v in pmc["kernels"].values()) * nb if pmc else None
        achieved = b_alg * nb / (gpu_ms_step * 1e-3) / 1e9
        line = {
            "metric": "stereo pairs/s at 1920x1080 numDisparities=128",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": agg["seconds"] / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": "cv2.StereoSGBM-equivalent disparity of %dx%d rectified %s pairs, "
                                   "numDisparities=%d blockSize=%d mode=%s, median3 on, speckle off"
                                   % (a.width, a.height, "RGB" if a.channels == 3 else "gray", a.disparities,
                                      a.block, "MODE_HH(8 paths)" if a.mode == "hh" else "MODE_SGBM(5 paths)"),
                       "pairs_per_gpu_per_step": a.batch, "global_pairs_per_step": world * a.batch,
                       "batches_in_flight_per_gpu": nfl, "host_cpus_bound": cpus_near_gpu,
                       "parallelism": "pairs sharded over %d GPU(s), no data-path collective" % world},
            # Headline = SURVEY section 8(d): B_alg x pairs per step / time of one step in the timed region, against
            # the 8 TB/s HBM peak.  `kernels` / `dominant_kernel` characterise every kernel ALONE on the GPU (hipEvents
            # on the compute stream, a few single-stream steps before the timed region; `kernel_ms_per_step` = their
            # sum): the kernel with the largest launch duration, priced with ITS algorithmic bytes.  Every number can
            # be recomputed from profiles/r02_*kernel_stats.csv and profiles/r02_pmc_traffic*.json.
            "roofline": {
                "bound": "hbm", "scope": "whole step (all kernels of one batch)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "algorithmic_bytes_per_pair": b_alg, "algorithmic_bytes_per_launch": float(b_alg * nb),
                "gpu_ms_per_step": gpu_ms_step, "kernel_ms_per_step": kernel_ms_step,
                "traffic": traffic_total,
                "traffic_ratio": (traffic_total / (b_alg * nb)) if traffic_total else None,
                "traffic_source": (PMC_PROFILE % ("_hh" if a.mode == "hh" else "")) + " (bytes per pair per launch x pairs per launch)" if pmc else None,
                "dominant_kernel": dict(kernels[dom], stage=dom) if dom else None,
                "kernels": kernels,
            },
        }
        if also:
            line["also"] = also
        if rccl:
            line["rccl"] = rccl
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a, params)
        print(json.dumps(line))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
