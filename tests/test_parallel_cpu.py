"""world_size-2 gloo test of the multi-GPU layer (runs on CPU): the one-time table broadcast and the
shard arithmetic bench.py / a multi-GPU driver rely on.  No per-pair collective exists to test."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import calibrating_amd as ca
        from calibrating_amd import synthetic
        from calibrating_amd.parallel_pairs import aggregate, broadcast_tables, ranks_agree, shard_range, timed_steps
        bundle = None
        if rank == 0:  # ONLY rank 0 ever sees the rig record
            src = ca.Stereo.load(synthetic.rig(160, 120))
            src.set_stereo_matching(object(), max_depth=3.5)
            bundle = src.table_bundle()
        tabs = broadcast_tables(bundle, torch.device("cpu"), src=0)
        # worker-rank side of the broadcast: a rig from the bundle ALONE (6 maps + mask + 64 doubles)
        st = ca.Stereo.from_bundle(tabs, torch.device("cpu"))
        st.set_stereo_matching(object(), max_depth=3.5)
        ok = st._tables(torch.device("cpu"))["map2y"] is tabs["map2y"]
        ok = ok and st._unrectify_tables(torch.device("cpu"))[0] is tabs["unrect_mapx"]
        ok = ok and sorted(tabs) == sorted(["map1x", "map1y", "map2x", "map2y", "mask", "unrect_mapx", "unrect_mapy", "params"])
        nbytes = sum(t.numel() * t.element_size() for t in tabs.values())
        ok = ok and nbytes == 160 * 120 * (6 * 4 + 1) + 64 * 8
        # what the bundle-built rig derives must equal what a rig loaded from the record derives -- checked on every
        # rank against a local rebuild that is used for NOTHING else
        ref = ca.Stereo.load(synthetic.rig(160, 120))
        ref.set_stereo_matching(object(), max_depth=3.5)
        rb = ref.table_bundle()
        ok = ok and all(np.array_equal(tabs[k].numpy(), rb[k]) for k in rb)
        ok = ok and st.min_disparity == ref.min_disparity and st.baseline == ref.baseline and tuple(st.xy) == tuple(ref.xy)
        ok = ok and all(np.array_equal(getattr(st, k), getattr(ref, k)) for k in ("K", "R1", "R2", "t"))
        ok = ok and np.array_equal(st.cam1.K, ref.cam1.K) and np.array_equal(st.cam1.D, ref.cam1.D) \
            and tuple(st.cam1.xy) == tuple(ref.cam1.xy) and tuple(st.cam2.xy) == tuple(ref.cam2.xy)
        ok = ok and np.array_equal(st.rectify_valid_mask1, ref.rectify_valid_mask1)  # host views of the installed tables
        ok = ok and np.array_equal(st.undistort_rectify_map2[1], ref.undistort_rectify_map2[1])
        d = np.float32([[0.0, 3.5, 40.0], [7.25, 0.5, 12.0]])
        ok = ok and np.array_equal(st.disparity_to_depth(d.copy()), ref.disparity_to_depth(d.copy()))
        lo, hi = shard_range(11, world, rank)
        thr = aggregate(hi - lo, 1.0 + rank, -5 - rank, torch.device("cpu"), distributed=True)
        total, tmax = thr["total_pairs"], thr["seconds"]
        assert thr["checksum"] == sum(-5 - r for r in range(world))  # negative checksums survive as they are
        # the code path bench.py times and aggregates with (same functions, gloo instead of RCCL)
        calls = []
        dt = timed_steps(lambda: calls.append(1), steps=3, warmup=2, synchronize=None, distributed=True)
        big = (1 << 40) + 12345 + rank  # a checksum beyond float32 / int32 range must survive the reduction
        agg = aggregate(hi - lo, dt, big, torch.device("cpu"), distributed=True)
        same = ranks_agree(777, torch.device("cpu"))
        diff = ranks_agree(777 + rank, torch.device("cpu"))
        q.put((rank, ok, lo, hi, total, tmax, len(calls), agg["total_pairs"], agg["checksum"], agg["seconds"] >= dt,
               same, diff))
    finally:
        dist.destroy_process_group()


def test_table_broadcast_and_sharding_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "broadcast tables differ from the source"
    assert (res[0][2], res[0][3], res[1][2], res[1][3]) == (0, 6, 6, 11)
    assert res[0][4] == 11 and res[0][5] == 2.0
    for r in res:
        assert r[6] == 5                                   # 2 warm-up + 3 timed calls
        assert r[7] == 11                                  # pairs over both ranks
        assert r[8] == 2 * ((1 << 40) + 12345) + 1         # checksum of checksums, exact
        assert r[9] and r[10] and not r[11]                # MAX of seconds; agreement detector both ways


def test_bench_main_runs_end_to_end_with_8_ranks_on_cpu():
    """Config C3's harness (BASELINE.json configs[2]: 512 pairs over 8 GPUs) without GPUs: `bench.py --gpus 8 --backend
    gloo --stub-compute` runs main() itself -- self-launch under torch.distributed.run, shard arithmetic, process
    group, table broadcast and install, warm-up, the timed region with its barriers, the all_gather reductions, the
    agreement check, the JSON line -- with CPU tensors and a no-op matcher.  The first real N > 1 run on RCCL is the
    driver's; this keeps everything around the kernels from failing there for the first time."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1",
                        "--backend", "gloo", "--stub-compute", "--width", "96", "--height", "64", "--disparities", "16"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_pairs_per_step"] == 512 and d["config"]["pairs_per_gpu_per_step"] == 64
    assert d["data"] == "stub" and "stub" in d
    r = d["rccl"]
    assert r["world_size"] == 8 and r["ranks_agree"] and len(r["per_rank"]) == 8
    assert all(row["pairs"] == 64 * 3 for row in r["per_rank"])
    assert abs(d["value"] - 8 * 64 * 3 / max(row["seconds"] for row in r["per_rank"])) < 1e-6 * d["value"]
    assert len({row["disparity_checksum"] for row in r["per_rank"]}) == 8  # every rank worked on its own shard


def test_bench_refuses_more_gpus_than_visible():
    """`bench.py --gpus 2` without that many GPUs must exit non-zero with a message, not report n_gpus=1."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert p.returncode != 0
    assert "GPU(s) visible" in p.stderr
    assert not p.stdout.strip()
