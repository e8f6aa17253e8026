#!/usr/bin/env python3
"""bench.py -- stereo pairs/s at 1920x1080, numDisparities=128 on MI355X (BASELINE.json metric).

A step = one pass of the SGBM hot path (BT cost volume -> path aggregation -> WTA / uniqueness /
LR check -> median) over one batch of synthetic rectified pairs that are already resident in HBM.
Independent pairs shard across ranks with no data-path collective ("scaling": "weak": every rank
processes its own batch); the only RCCL traffic is the one-time broadcast of the rig's remap tables
and the end-of-run timing reduction.  Launch: python bench.py [--gpus N --steps K --warmup W]; for
N > 1 through torch.distributed.run, one rank per GPU.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="pairs per GPU per step")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--disparities", type=int, default=128)
    ap.add_argument("--block", type=int, default=5)
    ap.add_argument("--channels", type=int, default=3, help="3 = RGB as the reference feeds SGBM, 1 = gray")
    ap.add_argument("--mode", default="hh", choices=["sgbm", "hh"],
                    help="hh = 8-path MODE_HH (north star), sgbm = 5-path MODE_SGBM (the reference's call)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs in the CPU baseline sample (0 = one per core)")
    return ap.parse_args()


def sgbm_params(a):
    cn, bs = a.channels, a.block
    return dict(minDisparity=0, numDisparities=a.disparities, blockSize=bs, P1=8 * cn * bs * bs,
                P2=32 * cn * bs * bs, disp12MaxDiff=1, preFilterCap=0, uniquenessRatio=10,
                speckleWindowSize=0, speckleRange=0, mode=1 if a.mode == "hh" else 0)


def algorithmic_bytes_per_pair(W, H, D, cn, minD=0):
    """B_alg = 2*V + IO, V = H*W1*D*2, IO = 2*H*W*cn + 2*H*W."""
    maxD = minD + D
    W1 = (W + min(minD, 0)) - max(maxD, 0)
    V = H * W1 * D * 2
    return 2 * V + 2 * H * W * cn + 2 * H * W, V


def cpu_baseline(a, params):
    """The CPU oracle (scalar C port of cv2.StereoSGBM, oracle/sgbm_ref.c) on this host's cores."""
    import oracle
    from calibrating_amd import synthetic
    oracle.build()
    threads = min(os.cpu_count() or 1, 32)
    n = a.cpu_pairs or threads
    # bounded sample: full-width strips of a quarter of the rows (SGBM cost is linear in rows), one
    # strip per thread; scaled back to whole pairs below
    frac = 4
    hs = max(a.height // frac, 16)
    base_l, base_r = synthetic.rectified_pair(seed=1234, H=hs, W=a.width, D=a.disparities, cn=a.channels)
    lefts, rights = [], []
    for i in range(n):  # distinct strips: vertical rolls of one generated strip
        lefts.append(np.roll(base_l, 17 * i, axis=0))
        rights.append(np.roll(base_r, 17 * i, axis=0))
    L, R = np.stack(lefts), np.stack(rights)
    t0 = time.perf_counter()
    oracle.sgbm_compute_batch(L, R, nthreads=threads, **params)
    dt = time.perf_counter() - t0
    pairs = n * hs / a.height
    return dict(value=pairs / dt, unit="pairs/s", cores=threads, kind="port",
                sample="%d strips of %dx%d (= %.2f pairs of %dx%d) D=%d cn=%d mode=%s, scalar C port "
                       "oracle/sgbm_ref.c, %d OpenMP threads (one strip each), %.1f s"
                       % (n, a.width, hs, pairs, a.width, a.height, a.disparities, a.channels, a.mode,
                          threads, dt))


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        a.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import calibrating_amd as ca
    from calibrating_amd import synthetic
    from calibrating_amd.parallel_pairs import broadcast_tables, shard_range

    params = sgbm_params(a)
    # one-time table broadcast (rank 0 owns the rig; every rank needs maps + mask for get_depth)
    bundle = None
    if rank == 0:
        bundle = ca.Stereo.load(synthetic.rig(a.width, a.height)).table_bundle()
    tables = broadcast_tables(bundle, dev, src=0) if distributed else None
    del tables

    # this rank's shard of the global pair list: pairs [lo, hi) of world*batch
    lo, hi = shard_range(world * a.batch, world, rank)
    nb = hi - lo
    left, right = synthetic.rectified_batch_torch(1234 + rank, nb, a.height, a.width, a.disparities,
                                                  a.channels, dev)
    matcher = ca.StereoSGBM_create(**params)
    matcher.set_profiling(True)
    out = torch.empty((nb, a.height, a.width), dtype=torch.int16, device=dev)

    def step():
        matcher.compute(left, right, out=out)

    for _ in range(a.warmup):
        step()
    stage_ms = {}
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        for k, v in matcher.stage_times_ms().items():  # hipEvents on the compute stream
            stage_ms[k] = stage_ms.get(k, 0.0) + v
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cs = torch.tensor([float(out.to(torch.int64).sum().item())], dtype=torch.float64, device=dev)
        dist.all_reduce(cs)  # checksum of checksums: touches every rank's result
    total_pairs = world * a.batch * a.steps
    value = total_pairs / dt

    if rank == 0:
        b_alg, V = algorithmic_bytes_per_pair(a.width, a.height, a.disparities, a.channels)
        npaths = 8 if a.mode == "hh" else 5
        gpu_ms_step = sum(stage_ms.values()) / a.steps
        achieved = b_alg * nb / (gpu_ms_step * 1e-3) / 1e9
        scan_ms_launch = stage_ms.get("scan", 0.0) / a.steps / npaths
        scan_bytes_launch = (3 * npaths - 1) / npaths * V * nb
        line = {
            "metric": "stereo pairs/s at 1920x1080 numDisparities=128",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int16", "data": "synthetic",
            "config": {"workload": "cv2.StereoSGBM-equivalent disparity of %dx%d rectified %s pairs, "
                                   "numDisparities=%d blockSize=%d mode=%s, median3 on, speckle off"
                                   % (a.width, a.height, "RGB" if a.channels == 3 else "gray", a.disparities,
                                      a.block, "MODE_HH(8 paths)" if a.mode == "hh" else "MODE_SGBM(5 paths)"),
                       "pairs_per_gpu_per_step": a.batch, "global_pairs_per_step": world * a.batch,
                       "parallelism": "pairs sharded over %d GPU(s), no data-path collective" % world},
            "roofline": {
                "bound": "hbm", "kernel": "SGBM pipeline (all launches of one step)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "algorithmic_bytes_per_pair": b_alg, "gpu_ms_per_step": gpu_ms_step,
                "stage_ms_per_step": {k: v / a.steps for k, v in stage_ms.items()},
                "dominant_kernel": {"name": "k_scan (one aggregation direction)",
                                    "launches_per_step": npaths, "avg_ms_per_launch": scan_ms_launch,
                                    "algorithmic_bytes_per_launch": scan_bytes_launch,
                                    "achieved_GBps": scan_bytes_launch / (scan_ms_launch * 1e-3) / 1e9
                                    if scan_ms_launch > 0 else None},
            },
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a, params)
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
