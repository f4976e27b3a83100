"""RCCL (xGMI) exchange of Y shards over the C ABI -- no torch needed.

One process per GPU.  Rank 0 creates the RCCL unique id and hands its 128 bytes
to the other ranks over a plain TCP socket on MASTER_ADDR:MASTER_PORT+port_offset
(the variables torch.distributed.run exports); then every rank builds its
communicator.  `torch.distributed` users can pass the id themselves (`Comm(rank,
world, comm_id=...)`).
"""
import ctypes
import os
import socket
import time


from . import _lib
from .device import DeviceArray, dtype_code, _stream_handle

ID_BYTES = 128


def unique_id():
    buf = ctypes.create_string_buffer(ID_BYTES)
    _lib.call("smm_comm_unique_id", buf)
    return buf.raw


def exchange_id(rank, world, addr=None, port=None, timeout=120.0):
    """Rank 0 serves the id to world-1 peers; the others fetch it."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port or int(os.environ.get("MASTER_PORT", "29500")) + 23)
    if world == 1:
        return unique_id()
    if rank == 0:
        cid = unique_id()
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world)
            srv.settimeout(timeout)
            for _ in range(world - 1):
                conn, _peer = srv.accept()
                with conn:
                    conn.sendall(cid)
        return cid
    deadline = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as c:
                data = b""
                while len(data) < ID_BYTES:
                    chunk = c.recv(ID_BYTES - len(data))
                    if not chunk:
                        break
                    data += chunk
            if len(data) == ID_BYTES:
                return data
        except OSError:
            pass
        if time.time() > deadline:
            raise TimeoutError("could not fetch the RCCL unique id from rank 0")
        time.sleep(0.2)


class Comm:
    """RCCL communicator of this process (call `smmregrid_amd.device.set_device(local_rank)` first)."""

    def __init__(self, rank, world, comm_id=None):
        self.rank, self.world = int(rank), int(world)
        if comm_id is None:
            comm_id = exchange_id(self.rank, self.world)
        if len(comm_id) != ID_BYTES:
            raise ValueError("RCCL unique id must be 128 bytes")
        h = ctypes.c_void_p()
        _lib.call("smm_comm_create", ctypes.create_string_buffer(comm_id, ID_BYTES), self.world, self.rank,
                  ctypes.byref(h))
        self.handle = h

    def gather(self, shard, root=0, out=None, stream=None):
        """Gather equal-sized shards (DeviceArray) to `root`; returns the (world, *shard.shape)
        DeviceArray on root, None elsewhere."""
        if self.rank == root and out is None:
            out = DeviceArray((self.world,) + shard.shape, shard.dtype)
        _lib.call("smm_comm_gather", self.handle, ctypes.c_void_p(shard.ptr),
                  ctypes.c_void_p(out.ptr) if out is not None else None, shard.size,
                  dtype_code(shard.dtype), int(root), _stream_handle(stream))
        return out if self.rank == root else None

    def allgather(self, shard, out=None, stream=None):
        if out is None:
            out = DeviceArray((self.world,) + shard.shape, shard.dtype)
        _lib.call("smm_comm_allgather", self.handle, ctypes.c_void_p(shard.ptr), ctypes.c_void_p(out.ptr),
                  shard.size, dtype_code(shard.dtype), _stream_handle(stream))
        return out

    def close(self):
        if getattr(self, "handle", None):
            _lib.call("smm_comm_destroy", self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
