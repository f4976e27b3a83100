"""Multi-process plumbing of the batch-sharded regrid -- no torch needed.

Two pieces, one process per GPU:

* `HostRendezvous`: the CONTROL plane.  Rank 0 listens on MASTER_ADDR:MASTER_PORT+23 (the variables
  ``torch.distributed.run`` and ``bench.py``'s own launcher export; ``SMM_RDV_PORT`` names another port),
  every other rank keeps one TCP connection to it.  Barriers, the max-over-ranks of a timing and the hand-over of the RCCL unique id
  are small host-side all-gathers over those sockets -- nothing of it touches a GPU, so it runs (and
  is tested) on any host.
* `Comm`: the DATA plane, RCCL over xGMI behind the C ABI (`smm_comm_*`): the gather / all-gather
  of the Y shards, device to device.  `Comm.gather_rows` is the asynchronous tile gather
  `smmregrid_amd.distributed.TiledRingGather` drives.

`torch.distributed` users can pass their own unique id (`Comm(rank, world, comm_id=...)`).
"""
import ctypes
import os
import socket
import struct
import time

from . import _lib
from .device import DeviceArray, Event, Stream, dtype_code, _stream_handle

ID_BYTES = 128
PORT_OFFSET = 23        # rendezvous port = MASTER_PORT + 23 (MASTER_PORT itself may belong to a torch store)


def unique_id():
    buf = ctypes.create_string_buffer(ID_BYTES)
    _lib.call("smm_comm_unique_id", buf)
    return buf.raw


def _recv_exact(sock, n):
    data = bytearray()
    while len(data) < n:
        chunk = sock.recv(n - len(data))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous connection")
        data += chunk
    return bytes(data)


def _send_msg(sock, payload):
    sock.sendall(struct.pack("<q", len(payload)) + payload)


def _recv_msg(sock):
    (n,) = struct.unpack("<q", _recv_exact(sock, 8))
    return _recv_exact(sock, n) if n else b""


class HostRendezvous:
    """Host-side all-gather / barrier / max over the ranks of one job (TCP star around rank 0).

    Every collective call must be made by all ranks in the same order.  Socket timeouts turn a
    missing rank into a `TimeoutError` instead of a hang."""

    def __init__(self, rank, world, addr=None, port=None, timeout=300.0):
        self.rank, self.world = int(rank), int(world)
        if self.world <= 0 or not 0 <= self.rank < self.world:
            raise ValueError("bad rank / world size")
        self.addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        if port is None:   # a launcher may name a port of its own (bench.py does); else MASTER_PORT + 23
            port = os.environ.get("SMM_RDV_PORT") or int(os.environ.get("MASTER_PORT", "29500")) + PORT_OFFSET
        self.port = int(port)
        self.timeout = float(timeout)
        self.peers = {}          # rank 0: rank -> socket
        self.sock = None         # other ranks: the connection to rank 0
        self.srv = None
        if self.world == 1:
            return
        if self.rank == 0:
            self.srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            self.srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            self.srv.bind((self.addr, self.port))
            self.srv.listen(self.world)
            self.srv.settimeout(self.timeout)
            while len(self.peers) < self.world - 1:
                try:
                    conn, _peer = self.srv.accept()
                except socket.timeout:
                    raise TimeoutError(f"rendezvous: {len(self.peers) + 1} of {self.world} ranks showed up") from None
                conn.settimeout(self.timeout)
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                (r,) = struct.unpack("<i", _recv_exact(conn, 4))
                if not 0 < r < self.world or r in self.peers:
                    conn.close()
                    raise ConnectionError(f"rendezvous: unexpected rank {r}")
                self.peers[r] = conn
        else:
            deadline = time.time() + self.timeout
            while True:
                try:
                    self.sock = socket.create_connection((self.addr, self.port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise TimeoutError(f"rendezvous: rank 0 not reachable at {self.addr}:{self.port}") from None
                    time.sleep(0.1)
            self.sock.settimeout(self.timeout)
            self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self.sock.sendall(struct.pack("<i", self.rank))

    @property
    def n_connected(self):
        """Ranks that really joined (rank 0 counts its accepted peers; the others learn it from
        the first all-gather)."""
        return self.world if self.world == 1 or self.rank else len(self.peers) + 1

    def allgather(self, payload=b""):
        """Every rank's payload (bytes), in rank order, on every rank."""
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        try:
            if self.rank == 0:
                parts = [payload] + [_recv_msg(self.peers[r]) for r in range(1, self.world)]
                blob = b"".join(struct.pack("<q", len(p)) + p for p in parts)
                for r in range(1, self.world):
                    _send_msg(self.peers[r], blob)
                return parts
            _send_msg(self.sock, payload)
            blob = _recv_msg(self.sock)
        except socket.timeout:
            raise TimeoutError("rendezvous: a rank did not reach the collective in time") from None
        parts, pos = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<q", blob, pos)
            parts.append(blob[pos + 8:pos + 8 + n])
            pos += 8 + n
        return parts

    def barrier(self):
        self.allgather(b"")

    def max(self, value):
        return max(struct.unpack("<d", p)[0] for p in self.allgather(struct.pack("<d", float(value))))

    def bcast(self, payload=None, root=0):
        """`payload` of `root` on every rank."""
        return self.allgather(payload if self.rank == root and payload is not None else b"")[root]

    def close(self):
        for s in list(self.peers.values()) + [self.sock, self.srv]:
            try:
                if s is not None:
                    s.close()
            except OSError:
                pass
        self.peers, self.sock, self.srv = {}, None, None

    def __del__(self):
        self.close()


def exchange_id(rank, world, addr=None, port=None, timeout=120.0, rendezvous=None):
    """The RCCL unique id of rank 0 on every rank: created by rank 0, handed over through
    `rendezvous` (or a rendezvous opened for this one exchange)."""
    if world == 1:
        return unique_id()
    own = rendezvous is None
    if own:
        rendezvous = HostRendezvous(rank, world, addr=addr, port=port, timeout=timeout)
    try:
        cid = rendezvous.bcast(unique_id() if rank == 0 else None)
    finally:
        if own:
            rendezvous.close()
    if len(cid) != ID_BYTES:
        raise ValueError("RCCL unique id must be 128 bytes")
    return cid


class _EventWork:
    """Handle of one asynchronous tile gather: wait() blocks the host until it has landed."""

    def __init__(self, event, parts, keep=None):
        self.event, self.parts, self.keep = event, parts, keep   # `keep`: events the queued work still refers to

    def wait(self):
        self.event.synchronize()
        return self.parts


class Comm:
    """RCCL communicator of this process (call `smmregrid_amd.device.set_device(local_rank)` first)."""

    def __init__(self, rank, world, comm_id=None, rendezvous=None):
        self.rank, self.world = int(rank), int(world)
        if comm_id is None:
            comm_id = exchange_id(self.rank, self.world, rendezvous=rendezvous)
        if len(comm_id) != ID_BYTES:
            raise ValueError("RCCL unique id must be 128 bytes")
        h = ctypes.c_void_p()
        _lib.call("smm_comm_create", ctypes.create_string_buffer(comm_id, ID_BYTES), self.world, self.rank,
                  ctypes.byref(h))
        self.handle = h
        self._stream = None

    # ---- blocking-free collectives on a caller-chosen stream -------------------------------
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

    # ---- the interface smmregrid_amd.distributed drives (shards stay in HBM) ----------------
    def alloc(self, shape, dtype):
        return DeviceArray(tuple(shape), dtype)

    def alloc_slot(self, shard, rows):
        """Receive buffer of one ring slot on the root: `rows` rows of every rank."""
        return DeviceArray((self.world, int(rows)) + tuple(shard.shape[1:]), shard.dtype)

    def rows(self, arr, r0, r1):
        return arr.rows(r0, r1)

    def to_host(self, arr):
        return arr.to_host()

    def gather_rows(self, shard, r0, r1, slot, root=0, compute_stream=None):
        """Start the gather of rows [r0, r1) of every rank's `shard` into `slot` (root only; None
        elsewhere) on the communication stream, behind everything queued so far on
        `compute_stream` (default: the null stream) -- so it overlaps whatever is launched next."""
        if self._stream is None:
            self._stream = Stream()
        ready, done = Event(), Event()
        ready.record(compute_stream)
        self._stream.wait_event(ready)
        view = shard.rows(r0, r1)
        n = r1 - r0
        out = parts = None
        if self.rank == root:
            # RCCL packs equal counts back to back: a short last tile fills the head of the slot
            out = DeviceArray((self.world, n) + tuple(shard.shape[1:]), shard.dtype, ptr=slot.ptr, base=slot)
            parts = [out.rows(r, r + 1).reshape((n,) + tuple(shard.shape[1:])) for r in range(self.world)]
        _lib.call("smm_comm_gather", self.handle, ctypes.c_void_p(view.ptr),
                  ctypes.c_void_p(out.ptr) if out is not None else None, view.size, dtype_code(view.dtype),
                  int(root), self._stream.handle)
        done.record(self._stream)
        return _EventWork(done, parts, keep=ready)

    def close(self):
        if getattr(self, "handle", None):
            _lib.call("smm_comm_destroy", self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
