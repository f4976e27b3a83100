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


def tile_bounds(n_rows, tiles):
    """Row tiles of the overlapped gather: at most `tiles` tiles of ceil(n_rows / tiles) rows, the
    last one short when the tile size does not divide the rows."""
    n_rows = int(n_rows)
    if n_rows <= 0:
        return []
    per = -(-n_rows // max(1, min(int(tiles), n_rows)))
    return [(r0, min(n_rows, r0 + per)) for r0 in range(0, n_rows, per)]


class TiledRingGather:
    """Gather of every rank's Y shard to `root`, tiled and overlapped with the compute.

    The shard (a torch tensor, rows first) is cut into row tiles; `gather_tile(k)` starts the
    asynchronous gather of tile k -- on the GPU box it travels over RCCL/xGMI while the kernel of
    tile k + 1 runs.  The root receives into a RING of `slots` tile buffers per rank, not into one
    buffer of the full size: the gathered Y of all ranks never has to fit on one GPU (BASELINE
    config 5: 211 GB).  A ring slot is reused only after the gather that last filled it has
    completed AND the consumer has drained it (`on_tile(k, parts)` is called on the root with the
    per-rank views of tile k exactly once, in tile order).

    Works with any torch.distributed backend (nccl == RCCL on the GPU box, gloo in CPU tests).
    """

    def __init__(self, dist, torch, shard, root=0, tiles=8, slots=2, group=None, on_tile=None):
        self.dist, self.torch, self.group = dist, torch, group
        self.shard, self.root, self.slots = shard, int(root), int(slots)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.tiles = tile_bounds(shard.shape[0], tiles)
        self.on_tile = on_tile
        per = max((r1 - r0 for r0, r1 in self.tiles), default=0)
        self.ring = None
        if self.rank == self.root:
            self.ring = [[torch.empty((per,) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
                          for _ in range(self.world)] for _ in range(self.slots)]
        self.pending = []          # (tile index, work handle), oldest first
        self.gathered_bytes = 0    # bytes received by the root from OTHER ranks (with_gather accounting)
        self.delivered = 0         # tiles handed to on_tile

    def _retire(self):
        k, work = self.pending.pop(0)
        work.wait()
        if self.rank == self.root:
            r0, r1 = self.tiles[k]
            parts = [buf[:r1 - r0] for buf in self.ring[k % self.slots]]
            self.gathered_bytes += (self.world - 1) * parts[0].numel() * parts[0].element_size()
            if self.on_tile is not None:
                self.on_tile(k, parts)
        self.delivered += 1

    def gather_tile(self, k):
        """Start the gather of tile k (call after the compute of tile k has been enqueued)."""
        r0, r1 = self.tiles[k]
        while len(self.pending) >= self.slots:      # the slot about to be reused must be drained
            self._retire()
        recv = [buf[:r1 - r0] for buf in self.ring[k % self.slots]] if self.rank == self.root else None
        work = self.dist.gather(self.shard[r0:r1], recv, dst=self.root, group=self.group, async_op=True)
        self.pending.append((k, work))

    def finish(self):
        while self.pending:
            self._retire()
