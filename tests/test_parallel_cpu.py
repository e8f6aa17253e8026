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
        if rank == 0:
            bundle = ca.Stereo.load(synthetic.rig(160, 120)).table_bundle()
        tabs = broadcast_tables(bundle, torch.device("cpu"), src=0)
        ref = ca.Stereo.load(synthetic.rig(160, 120)).table_bundle()  # every rank can rebuild it to compare
        ok = all(np.array_equal(tabs[k].numpy(), ref[k]) for k in ref)
        # worker-rank side of the broadcast: the bundle becomes the rig's device tables without a rebuild
        st = ca.Stereo.load(synthetic.rig(160, 120)).install_tables(tabs, torch.device("cpu"))
        ok = ok and st._tables(torch.device("cpu"))["map2y"] is tabs["map2y"]
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
