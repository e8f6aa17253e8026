#!/usr/bin/env python3
"""bench.py -- stereo pairs/s at 1920x1080, numDisparities=128 on MI355X (BASELINE.json metric).

A step = one pass of the SGBM hot path (BT cost volume -> path aggregation -> WTA / uniqueness /
LR check -> median) over one batch of synthetic rectified pairs that are already resident in HBM.
Independent pairs shard across ranks with no data-path collective ("scaling": "weak": every rank
processes its own batch of --batch pairs); the only RCCL traffic is the one-time broadcast of the rig's
remap tables and the end-of-run timing reduction.  Launch: python bench.py [--gpus N --steps K
--warmup W]; for N > 1 through torch.distributed.run, one rank per GPU.

Default workload = BASELINE.json configs[1]/[2]: 1920x1080 RGB pairs (the reference feeds RGB), D=128,
blockSize=5, cv2's default MODE_SGBM (5 paths: the mode the reference's cv2.StereoSGBM_create call
selects), 64 pairs per GPU per step (512 pairs / 8 GPUs).  --mode hh runs the 8-path MODE_HH; its
throughput is also reported in "also" on every default run.
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
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="pairs per GPU per step")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--disparities", type=int, default=128)
    ap.add_argument("--block", type=int, default=5)
    ap.add_argument("--channels", type=int, default=3, help="3 = RGB as the reference feeds SGBM, 1 = gray")
    ap.add_argument("--mode", default="sgbm", choices=["sgbm", "hh"],
                    help="sgbm = 5-path MODE_SGBM (the reference's call), hh = 8-path MODE_HH")
    ap.add_argument("--path", type=int, default=0, help="0 = fused band-wavefront passes, 1 = one scan per direction")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary (other mode / gray) measurements")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="strips in the CPU baseline sample (0 = one per thread)")
    return ap.parse_args()


def sgbm_params(a, mode=None):
    cn, bs = a.channels, a.block
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


def cpu_baseline(a, params):
    """The CPU oracle (scalar C port of cv2.StereoSGBM, oracle/sgbm_ref.c) on this host's cores."""
    import oracle
    from calibrating_amd import synthetic
    oracle.build()
    threads = min(os.cpu_count() or 1, 32)
    n = a.cpu_pairs or threads
    # bounded sample (~10-30 s of CPU work): full-width strips of half the rows (SGBM cost is linear in
    # rows), two strips per thread; scaled back to whole pairs below
    frac = 2
    hs = max(a.height // frac, 16)
    n = n * 2 if not a.cpu_pairs else n
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
                       "oracle/sgbm_ref.c, %d OpenMP threads across strips, %.1f s"
                       % (n, a.width, hs, pairs, a.width, a.height, a.disparities, a.channels, a.mode,
                          threads, dt))


def timed_steps(matcher, left, right, out, steps, warmup, barrier=None):
    """K timed steps bracketed by barrier + synchronize; returns (seconds, {stage: ms summed})."""
    import torch
    for _ in range(warmup):
        matcher.compute(left, right, out=out)
    stage_ms = {}
    if barrier:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        matcher.compute(left, right, out=out)
        for k, v in matcher.stage_times_ms().items():  # hipEvents on the compute stream
            stage_ms[k] = stage_ms.get(k, 0.0) + v
    torch.cuda.synchronize()
    if barrier:
        barrier()
    return time.perf_counter() - t0, stage_ms


def _kernel_line(ms, nbytes):
    return {"avg_ms_per_launch": ms, "algorithmic_bytes_per_launch": float(nbytes),
            "achieved": (nbytes / (ms * 1e-3) / 1e9) if ms > 0 else None, "unit": "GB/s"}


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        a.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # CAMD_BENCH_FORCE_DIST=1 exercises the RCCL path (init, table broadcast, barrier, all-reduce) with a
    # single rank, e.g. under `python -m torch.distributed.run --nproc-per-node 1`
    distributed = world > 1 or (os.environ.get("CAMD_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["NCCL_DEBUG"] = os.environ.get("CAMD_NCCL_DEBUG", "WARN")  # keep RCCL's banner off stdout
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import calibrating_amd as ca
    from calibrating_amd import synthetic
    from calibrating_amd.parallel_pairs import broadcast_tables, shard_range

    params = sgbm_params(a)
    # one-time table broadcast (rank 0 owns the rig; every rank needs maps + mask for get_depth)
    if distributed:
        bundle = ca.Stereo.load(synthetic.rig(a.width, a.height)).table_bundle() if rank == 0 else None
        tables = broadcast_tables(bundle, dev, src=0)
        del tables

    # this rank's shard of the global pair list: pairs [lo, hi) of world*batch
    lo, hi = shard_range(world * a.batch, world, rank)
    nb = hi - lo
    left, right = synthetic.rectified_batch_torch(1234 + rank, nb, a.height, a.width, a.disparities,
                                                  a.channels, dev)
    matcher = ca.StereoSGBM_create(**params)
    matcher.set_profiling(True)
    matcher.set_option("path", a.path)
    out = torch.empty((nb, a.height, a.width), dtype=torch.int16, device=dev)

    dt, stage_ms = timed_steps(matcher, left, right, out, a.steps, a.warmup, dist.barrier if distributed else None)
    matcher.status()  # raises if a device-side bounded wait timed out
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cs = torch.tensor([float(out.to(torch.int64).sum().item())], dtype=torch.float64, device=dev)
        dist.all_reduce(cs)  # checksum of checksums: touches every rank's result
    total_pairs = world * a.batch * a.steps
    value = total_pairs / dt

    also = {}
    if rank == 0 and world == 1 and not a.no_also:
        del matcher
        # the other aggregation mode on the same inputs, and the gray variant of the headline mode
        other = "hh" if a.mode == "sgbm" else "sgbm"
        m2 = ca.StereoSGBM_create(**sgbm_params(a, other))
        m2.set_profiling(True)
        m2.set_option("path", a.path)
        d2, _ = timed_steps(m2, left, right, out, a.steps, 1)
        also["%s_%s_pairs_per_s" % ("rgb" if a.channels == 3 else "gray", other)] = nb * a.steps / d2
        del m2
        if a.channels == 3:
            g = argparse.Namespace(**vars(a))
            g.channels = 1
            m3 = ca.StereoSGBM_create(**sgbm_params(g))
            m3.set_profiling(True)
            m3.set_option("path", a.path)
            gl, gr = left[..., 1].contiguous(), right[..., 1].contiguous()
            d3, _ = timed_steps(m3, gl, gr, out, a.steps, 1)
            also["gray_%s_pairs_per_s" % a.mode] = nb * a.steps / d3
            del m3, gl, gr

    if rank == 0:
        b_alg, V = algorithmic_bytes_per_pair(a.width, a.height, a.disparities, a.channels)
        gpu_ms_step = sum(stage_ms.values()) / a.steps
        achieved = b_alg * nb / (gpu_ms_step * 1e-3) / 1e9
        if a.path == 0:
            # fused band passes (both modes): pass 1 reads C, writes S; pass 2 reads C and S (+ WTA).  The two are
            # different kernels, timed separately; the roofline block prices the slower (dominant) one.
            ms1 = stage_ms.get("scan", 0.0) / a.steps
            ms2 = stage_ms.get("scan_last", 0.0) / a.steps
            first_dominant = ms1 >= ms2
            k_ms = ms1 if first_dominant else ms2
            npass = 1
            vols = 2
            kname = ("k_band first pass (four directions: C in, S out)" if first_dominant else
                     "k_band last pass (%s + WTA: C and S in)" % ("four directions" if a.mode == "hh" else "one direction"))
            other = {"kernel": "k_band last pass" if first_dominant else "k_band first pass",
                     "avg_ms_per_launch": ms2 if first_dominant else ms1,
                     "achieved": (2 * V * nb / ((ms2 if first_dominant else ms1) * 1e-3) / 1e9) if min(ms1, ms2) > 0 else None}
        else:
            npass = 8 if a.mode == "hh" else 5
            vols = 3 * npass - 1
            kname = "k_scan (one aggregation direction)"
            k_ms = (stage_ms.get("scan", 0.0) + stage_ms.get("scan_last", 0.0)) / a.steps / npass
            other = None
        k_bytes = vols / npass * V * nb
        # HBM traffic per launch from the committed PMC profile of the same kernels (separate rocprofv3
        # --pmc passes, 2*FETCH_SIZE + WRITE_SIZE per the gfx950 correction); null when no profile matches
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic%s.json" % ("_hh" if a.mode == "hh" else ""))
        if (a.path == 0 and a.channels == 3 and (a.width, a.height, a.disparities) == (1920, 1080, 128)
                and os.path.exists(pmc_path)):
            pmc = json.load(open(pmc_path))
            # the dominant kernel's entry: first pass = k_band<.., true, 0, ..>, last pass = k_band<.., false|true, 2, ..>
            want = ", true, 0," if first_dominant else ", 2,"
            per_pair = [v["hbm_bytes_per_pair"] for k, v in pmc["kernels"].items() if "k_band" in k and want in k]
            if per_pair:
                traffic = per_pair[0] * nb
                traffic_src = "profiles/%s (bytes per pair per launch x pairs per launch)" % os.path.basename(pmc_path)
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
                "bound": "hbm", "kernel": kname,
                "achieved": k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (k_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if k_ms > 0 else None,
                "traffic": traffic, "traffic_source": traffic_src,
                "launches_per_step": npass, "avg_ms_per_launch": k_ms, "other_band_pass": other,
                # the rest of the pipeline, same accounting (algorithmic HBM bytes per launch / measured time)
                "other_kernels": {
                    "k_hsum (BT cost + horizontal box sum; VALU-bound)": _kernel_line(
                        stage_ms.get("hsum", 0.0) / a.steps, (V + 2 * a.width * a.height * a.channels) * nb),
                    "k_vsum (vertical box sum)": _kernel_line(stage_ms.get("vsum", 0.0) / a.steps, 2 * V * nb),
                },
                "algorithmic_bytes_per_launch": k_bytes,
                "stage_ms_per_step": {k: v / a.steps for k, v in stage_ms.items()},
                "pipeline": {"algorithmic_bytes_per_pair": b_alg, "gpu_ms_per_step": gpu_ms_step,
                             "achieved": achieved, "frac": achieved / HBM_PEAK_GBS},
            },
        }
        if also:
            line["also"] = also
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(a, params)
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
