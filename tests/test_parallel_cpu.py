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
        from calibrating_amd.parallel_pairs import broadcast_tables, gather_throughput, shard_range
        bundle = None
        if rank == 0:
            bundle = ca.Stereo.load(synthetic.rig(160, 120)).table_bundle()
        tabs = broadcast_tables(bundle, torch.device("cpu"), src=0)
        ref = ca.Stereo.load(synthetic.rig(160, 120)).table_bundle()  # every rank can rebuild it to compare
        ok = all(np.array_equal(tabs[k].numpy(), ref[k]) for k in ref)
        lo, hi = shard_range(11, world, rank)
        total, tmax = gather_throughput(hi - lo, 1.0 + rank, torch.device("cpu"))
        q.put((rank, ok, lo, hi, total, tmax))
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
