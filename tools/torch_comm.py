"""torch.distributed adapter for the communicator interface of smmregrid_amd.distributed.

NOT part of the product (the product communicator is smmregrid_amd.comm.Comm, RCCL behind the C
ABI, no torch).  Used by the CPU tests (backend "gloo", world_size 2) to drive the sharding and
the tiled ring gather without a GPU, and by `bench.py --comm torch` (backend "nccl" == RCCL) as
the alternative plumbing.  Arrays are torch tensors; a `DeviceArray` handed in is wrapped through
`__cuda_array_interface__` without a copy.
"""
import numpy as np


class _Work:
    def __init__(self, work, parts):
        self.work, self.parts = work, parts

    def wait(self):
        self.work.wait()
        return self.parts


class TorchComm:
    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if device is None:
            device = (torch.device("cuda", torch.cuda.current_device())
                      if dist.get_backend(group) == "nccl" else torch.device("cpu"))
        self.device = device

    def _dtype(self, dtype):
        return {np.dtype(np.float64): self.torch.float64, np.dtype(np.float32): self.torch.float32}[np.dtype(dtype)]

    def tensor(self, arr):
        """torch view of a tensor / numpy array / DeviceArray (no copy)."""
        if isinstance(arr, self.torch.Tensor):
            return arr
        if isinstance(arr, np.ndarray):
            return self.torch.from_numpy(arr)
        return self.torch.as_tensor(arr, device=self.device)      # __cuda_array_interface__

    def alloc(self, shape, dtype):
        return self.torch.zeros(tuple(shape), dtype=self._dtype(dtype), device=self.device)

    def alloc_slot(self, shard, rows):
        shard = self.tensor(shard)
        return self.torch.empty((self.world, int(rows)) + tuple(shard.shape[1:]), dtype=shard.dtype,
                                device=shard.device)

    def rows(self, arr, r0, r1):
        return arr[r0:r1]

    def to_host(self, arr):
        return arr.cpu().numpy()

    def gather(self, shard, root=0, out=None):
        shard = self.tensor(shard)
        parts = None
        if self.rank == root:
            if out is None:
                out = self.torch.empty((self.world,) + tuple(shard.shape), dtype=shard.dtype, device=shard.device)
            parts = [out[r] for r in range(self.world)]
        self.dist.gather(shard, parts, dst=root, group=self.group)
        return out if self.rank == root else None

    def allgather(self, shard, out=None):
        shard = self.tensor(shard)
        if out is None:
            out = self.torch.empty((self.world,) + tuple(shard.shape), dtype=shard.dtype, device=shard.device)
        self.dist.all_gather([out[r] for r in range(self.world)], shard, group=self.group)
        return out

    def gather_rows(self, shard, r0, r1, slot, root=0):
        shard = self.tensor(shard)
        parts = [slot[r, :r1 - r0] for r in range(self.world)] if self.rank == root else None
        work = self.dist.gather(shard[r0:r1], parts, dst=root, group=self.group, async_op=True)
        return _Work(work, parts)
