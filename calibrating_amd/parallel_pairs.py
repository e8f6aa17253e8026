"""Multi-GPU layer of the stereo path: independent pairs shard across ranks, tables are broadcast once.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the
CPU tests).  There is no per-pair communication: a rig's remap tables / validity mask (~52 MB at
1080p) are broadcast from rank 0 once, then every rank runs ``Stereo.get_depth`` / ``StereoSGBM.compute``
on its contiguous shard of the pair list.
"""
import numpy as np


def shard_range(n_pairs, world_size, rank):
    """Contiguous shard [lo, hi) of pair indices for ``rank`` (pair i -> rank floor(i*G/N))."""
    lo = (rank * n_pairs + world_size - 1) // world_size
    hi = ((rank + 1) * n_pairs + world_size - 1) // world_size
    return lo, hi


def owner_of(pair_index, n_pairs, world_size):
    return pair_index * world_size // n_pairs


_BUNDLE_KEYS = ("map1x", "map1y", "map2x", "map2y", "mask")


def broadcast_tables(bundle, device, src=0):
    """Broadcast the table bundle of ``Stereo.table_bundle()`` from ``src`` to every rank.

    ``bundle`` is the dict on ``src`` and ignored (may be None) elsewhere.  Shapes travel first in one
    small int64 tensor, then one collective per table.  Returns a dict of tensors on ``device``.
    """
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    meta = torch.zeros(2, dtype=torch.int64, device=device)
    if rank == src:
        h, w = bundle["map1x"].shape
        meta[0], meta[1] = h, w
    dist.broadcast(meta, src=src)
    h, w = int(meta[0].item()), int(meta[1].item())
    out = {}
    for k in _BUNDLE_KEYS:
        dtype = torch.uint8 if k == "mask" else torch.float32
        if rank == src:
            t = torch.from_numpy(np.ascontiguousarray(bundle[k])).to(device)
            assert t.dtype == dtype and tuple(t.shape) == (h, w)
        else:
            t = torch.empty((h, w), dtype=dtype, device=device)
        dist.broadcast(t, src=src)
        out[k] = t
    return out


def gather_throughput(pairs_done, seconds, device):
    """all_gather of per-rank (pairs, seconds); returns (total_pairs, max_seconds)."""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([float(pairs_done), float(seconds)], dtype=torch.float64, device=device)
    allv = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(allv, mine)
    return sum(float(v[0]) for v in allv), max(float(v[1]) for v in allv)
