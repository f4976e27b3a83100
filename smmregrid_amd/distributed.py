"""Batch-axis sharding over the GPUs of one node.

Every batch row (one time x level slab) is an independent product against
read-only weights (regrid.py:529-541, :550), so ranks take contiguous blocks of
the row axis, each rank holds its own copy of the operator, and there is no
collective on the data path except the optional gather of the output shards.

One process per GPU.  This module is sharding arithmetic plus the gather
schedules over an INJECTED communicator object; it imports neither torch nor
the HIP library.  The product communicator is `smmregrid_amd.comm.Comm` (RCCL
over xGMI behind the C ABI: shards are `DeviceArray`s and never leave HBM
before the collective); tests inject a gloo adapter (`tools/torch_comm.py`, also behind
`bench.py --comm torch`) and the CPU oracle as the per-rank compute.

Communicator interface (duck-typed):

    comm.rank, comm.world
    comm.alloc(shape, dtype)               -> array of the communicator's kind
    comm.alloc_slot(shard, rows)           -> receive buffer for `rows` rows of every rank
    comm.rows(arr, r0, r1)                 -> view of rows [r0, r1)
    comm.to_host(arr)                      -> numpy array
    comm.gather(shard, root, out=None)     -> (world, *shard.shape) array on root, None elsewhere
    comm.allgather(shard, out=None)        -> (world, *shard.shape) array
    comm.gather_rows(shard, r0, r1, slot, root) -> work, with work.wait() -> per-rank views (root) / None
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


def regrid_sharded(x_rows, apply_fn, n_dst, comm, gather="root", root=0, dtype=np.float64):
    """Regrid batch rows (B, S) with the rows split over the ranks of `comm`.  Every rank
    passes the same full `x_rows` (or at least its own block of it) and gets back

      gather="root": the assembled (B, n_dst) numpy array on `root`, None elsewhere;
      gather="all" : the assembled array on every rank;
      gather="none": its own (rows, n_dst) shard, as the communicator's array kind.

    `apply_fn(rows_2d, out)` is the per-rank compute: it regrids `rows_2d` (this rank's block of
    `x_rows`) INTO `out`, a (rows, n_dst) view of a buffer of the communicator's kind -- for the
    product communicator a `DeviceArray`, so the shard is produced in HBM, travels device to
    device, and only the assembled result is copied to the host (once, on the receiving ranks).
    """
    if gather not in ("root", "all", "none"):
        raise ValueError("gather must be 'root', 'all' or 'none'")
    world, rank = comm.world, comm.rank
    n_rows = x_rows.shape[0]
    lo, hi = shard_bounds(n_rows, world, rank)
    per = -(-n_rows // world)                     # equal-sized shards for the collective
    pad = comm.alloc((max(per, 1), n_dst), dtype)
    if hi > lo:
        apply_fn(x_rows[lo:hi], comm.rows(pad, 0, hi - lo))
    if gather == "none":
        return comm.rows(pad, 0, hi - lo)      # a view: the caller owns its base (`.base.free()` for a DeviceArray)
    if gather == "all":
        parts = comm.allgather(pad)
    else:
        parts = comm.gather(pad, root)
    _release(pad)                               # device buffers are returned now, not at garbage collection
    if parts is None:
        return None
    host = comm.to_host(parts).reshape(world, max(per, 1), n_dst)
    _release(parts)
    out = np.empty((n_rows, n_dst), dtype=dtype)
    for r in range(world):
        a, b = shard_bounds(n_rows, world, r)
        if b > a:
            out[a:b] = host[r, :b - a]
    return out


def _release(arr):
    """Free a buffer of the communicator's kind if it has an explicit free (DeviceArray); numpy arrays and
    torch tensors are left to their own memory management."""
    free = getattr(arr, "free", None)
    if callable(free):
        free()


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

    The shard (rows first, the communicator's array kind) is cut into row tiles; `gather_tile(k)`
    starts the asynchronous gather of tile k -- on the GPU box it travels over RCCL/xGMI on the
    communication stream while the kernel of tile k + 1 runs.  The root receives into a RING of
    `slots` tile buffers, not into one buffer of the full size: the gathered Y of all ranks never
    has to fit on one GPU (BASELINE config 5: 211 GB).  A ring slot is reused only after the gather
    that last filled it has completed AND the consumer has drained it (`on_tile(k, parts)` is
    called on the root with the per-rank views of tile k exactly once, in tile order).
    """

    def __init__(self, comm, shard, root=0, tiles=8, slots=2, on_tile=None):
        self.comm, self.shard, self.root, self.slots = comm, shard, int(root), int(slots)
        self.rank, self.world = comm.rank, comm.world
        self.tiles = tile_bounds(shard.shape[0], tiles)
        self.on_tile = on_tile
        self.rows_per_slot = max((r1 - r0 for r0, r1 in self.tiles), default=0)
        self.ring = None
        if self.rank == self.root:
            self.ring = [comm.alloc_slot(shard, self.rows_per_slot) for _ in range(self.slots)]
        self.row_bytes = int(np.prod(shard.shape[1:], dtype=np.int64)) * np.dtype(shard_dtype(shard)).itemsize
        self.pending = []          # (tile index, work handle), oldest first
        self.gathered_bytes = 0    # bytes received by the root from OTHER ranks (with_gather accounting)
        self.delivered = 0         # tiles handed to on_tile

    def _retire(self):
        k, work = self.pending.pop(0)
        parts = work.wait()
        if self.rank == self.root:
            r0, r1 = self.tiles[k]
            self.gathered_bytes += (self.world - 1) * (r1 - r0) * self.row_bytes
            if self.on_tile is not None:
                self.on_tile(k, parts)
        self.delivered += 1

    def gather_tile(self, k):
        """Start the gather of tile k (call after the compute of tile k has been enqueued)."""
        r0, r1 = self.tiles[k]
        while len(self.pending) >= self.slots:      # the slot about to be reused must be drained
            self._retire()
        slot = self.ring[k % self.slots] if self.rank == self.root else None
        self.pending.append((k, self.comm.gather_rows(self.shard, r0, r1, slot, self.root)))

    def finish(self):
        while self.pending:
            self._retire()


def shard_dtype(shard):
    """numpy dtype of an array of any communicator's kind (DeviceArray / numpy: .dtype; torch
    tensors carry a torch dtype whose itemsize is what the byte accounting needs)."""
    dt = shard.dtype
    try:
        return np.dtype(dt)
    except TypeError:
        return np.dtype(f"V{shard.element_size()}")
