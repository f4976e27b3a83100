"""Batch-axis sharding over the GPUs of one node.

Every batch row (one time x level slab) is an independent product against
read-only weights (regrid.py:529-541, :550), so ranks take contiguous blocks of
the row axis, each rank holds its own copy of the operator, and there is no
collective on the data path except the optional gather of the output shards.

One process per GPU, launched with ``torch.distributed.run``; the process group
is plumbing only (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in
CPU tests).  The compute itself is injected as ``apply_fn`` so that CPU tests
can drive the sharding logic with the oracle while the product path passes the
HIP operator.
"""
import numpy as np


def shard_bounds(n_rows, world_size, rank):
    """Contiguous block of ceil(n_rows / world) rows per rank (last ranks may be short/empty)."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad rank / world size")
    per = -(-int(n_rows) // world_size)
    lo = min(n_rows, rank * per)
    hi = min(n_rows, lo + per)
    return lo, hi


def shard_sizes(n_rows, world_size):
    return [shard_bounds(n_rows, world_size, r)[1] - shard_bounds(n_rows, world_size, r)[0]
            for r in range(world_size)]


def regrid_sharded(x_rows, apply_fn, n_dst, group=None, gather="root", root=0):
    """Regrid a host array of batch rows (B, S) with the rows split over the ranks of
    `group`.  Every rank passes the same full `x_rows` (or at least its own block
    of it) and gets back

      gather="root": the assembled (B, n_dst) array on `root`, None elsewhere;
      gather="all" : the assembled array on every rank;
      gather="none": its own (rows, n_dst) shard.

    `apply_fn(rows_2d) -> (rows, n_dst) float64 ndarray` is the per-rank compute.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_rows = x_rows.shape[0]
    lo, hi = shard_bounds(n_rows, world, rank)
    mine = np.ascontiguousarray(apply_fn(x_rows[lo:hi]), dtype=np.float64).reshape(hi - lo, n_dst)
    if gather == "none":
        return mine

    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    per = -(-n_rows // world)
    pad = torch.zeros((per, n_dst), dtype=torch.float64, device=dev)   # equal-sized shards for the collective
    if hi > lo:
        pad[:hi - lo] = torch.from_numpy(mine).to(dev)
    if gather == "all":
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad, group=group)
    elif gather == "root":
        parts = [torch.empty_like(pad) for _ in range(world)] if rank == root else None
        dist.gather(pad, parts, dst=root, group=group)
        if rank != root:
            return None
    else:
        raise ValueError("gather must be 'root', 'all' or 'none'")
    out = np.empty((n_rows, n_dst), dtype=np.float64)
    for r, part in enumerate(parts):
        a, b = shard_bounds(n_rows, world, r)
        if b > a:
            out[a:b] = part[:b - a].cpu().numpy()
    return out
