"""Multi-GPU layer of the stereo path: independent pairs shard across ranks, tables are broadcast once.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the
CPU tests).  There is no per-pair communication: a rig's table bundle (six float32 maps, the validity
mask and a 64-double parameter block: ~52 MB at 1080p) is broadcast from rank 0 once and every other rank
builds its ``Stereo`` from it alone (``Stereo.from_bundle``: a worker never sees the rig record), then every rank runs ``Stereo.get_depth_batch`` / ``StereoSGBM.compute``
on its contiguous shard of the pair list.  ``bench.py`` is built from the
functions of this module, so the world_size-2 gloo test exercises the same code the RCCL run does.
"""
import time

import numpy as np


def shard_range(n_pairs, world_size, rank):
    """Contiguous shard [lo, hi) of pair indices for ``rank`` (pair i -> rank floor(i*G/N))."""
    lo = (rank * n_pairs + world_size - 1) // world_size
    hi = ((rank + 1) * n_pairs + world_size - 1) // world_size
    return lo, hi


def owner_of(pair_index, n_pairs, world_size):
    return pair_index * world_size // n_pairs


_BUNDLE_KEYS = ("map1x", "map1y", "map2x", "map2y", "mask", "unrect_mapx", "unrect_mapy", "params")


def broadcast_tables(bundle, device, src=0):
    """Broadcast the table bundle of ``Stereo.table_bundle()`` from ``src`` to every rank.

    ``bundle`` is the dict on ``src`` and ignored (may be None) elsewhere.  Shapes travel first in one
    small int64 tensor, then one collective per table.  Returns a dict of tensors on ``device``
    (feed it to ``Stereo.from_bundle``, or to ``Stereo.install_tables`` of a rig built from the record).
    """
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    # shapes first: (h, w) of the rectified frame and of camera 1's frame (the unrectify maps live there)
    meta = torch.zeros(4, dtype=torch.int64, device=device)
    if rank == src:
        meta[0], meta[1] = bundle["map1x"].shape
        meta[2], meta[3] = bundle["unrect_mapx"].shape
    dist.broadcast(meta, src=src)
    hw, hw1 = (int(meta[0].item()), int(meta[1].item())), (int(meta[2].item()), int(meta[3].item()))
    out = {}
    for k in _BUNDLE_KEYS:
        dtype = torch.uint8 if k == "mask" else (torch.float64 if k == "params" else torch.float32)
        shape = (64,) if k == "params" else (hw1 if k.startswith("unrect_") else hw)
        if rank == src:
            t = torch.from_numpy(np.ascontiguousarray(bundle[k])).to(device)
            assert t.dtype == dtype and tuple(t.shape) == shape, (k, t.dtype, tuple(t.shape))
        else:
            t = torch.empty(shape, dtype=dtype, device=device)
        dist.broadcast(t, src=src)
        out[k] = t
    return out


def timed_steps(step, steps, warmup, synchronize=None, distributed=False):
    """``warmup`` untimed + exactly ``steps`` timed calls of ``step()``, bracketed by barrier +
    ``synchronize()`` on both sides.  Returns this rank's seconds (reduce with ``aggregate``)."""
    sync = synchronize or (lambda: None)
    if distributed:
        import torch.distributed as dist
        barrier = dist.barrier
    else:
        barrier = lambda: None  # noqa: E731
    for _ in range(warmup):
        step()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    barrier()
    return time.perf_counter() - t0


def aggregate(pairs_done, seconds, checksum, device, distributed=True):
    """End-of-run reduction: total pairs over ranks, MAX of the seconds, and the checksum of checksums
    (sum over ranks of each rank's int64 result checksum -- touches every rank's output).

    Returns dict(total_pairs, seconds, checksum, per_rank=[(pairs, seconds, checksum), ...])."""
    if not distributed:
        return dict(total_pairs=int(pairs_done), seconds=float(seconds), checksum=int(checksum),
                    per_rank=[(int(pairs_done), float(seconds), int(checksum))])
    import torch
    import torch.distributed as dist
    n = dist.get_world_size()
    mine = torch.tensor([float(pairs_done), float(seconds)], dtype=torch.float64, device=device)
    allv = [torch.zeros_like(mine) for _ in range(n)]
    dist.all_gather(allv, mine)
    # the checksum travels as the int64 it is (sums of int16 disparities with many -16 pixels are negative)
    cs = torch.tensor([int(checksum)], dtype=torch.int64, device=device)
    allc = [torch.zeros_like(cs) for _ in range(n)]
    dist.all_gather(allc, cs)
    rows = [(int(v[0].item()), float(v[1].item()), int(c[0].item())) for v, c in zip(allv, allc)]
    return dict(total_pairs=sum(r[0] for r in rows), seconds=max(r[1] for r in rows),
                checksum=sum(r[2] for r in rows), per_rank=rows)


def ranks_agree(value, device):
    """True on every rank iff all ranks hold the same int64 ``value`` (MIN == MAX all-reduce)."""
    import torch
    import torch.distributed as dist
    v = int(value)
    t = torch.tensor([v, -v], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t[0].item()) == -int(t[1].item())
