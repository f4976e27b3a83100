"""HBM buffers, streams and events over the C ABI (no torch, no cupy)."""
import ctypes

import numpy as np

from . import _lib

_DTYPE_CODE = {np.dtype(np.float32): _lib.SMM_F32, np.dtype(np.float64): _lib.SMM_F64}


def dtype_code(dtype):
    try:
        return _DTYPE_CODE[np.dtype(dtype)]
    except KeyError:
        raise TypeError(f"field dtype {dtype} not supported on device (float32/float64 only)")


def device_count():
    return _lib.device_count()


def set_device(device):
    _lib.call("smm_set_device", int(device))


def current_device():
    d = ctypes.c_int(0)
    _lib.call("smm_get_device", ctypes.byref(d))
    return d.value


def device_name(device=0):
    buf = ctypes.create_string_buffer(256)
    _lib.call("smm_device_name", int(device), buf, 256)
    return buf.value.decode().strip()        # boxes without a marketing name report " (gfx950..., 256 CUs)"


def mem_info():
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _lib.call("smm_mem_info", ctypes.byref(f), ctypes.byref(t))
    return f.value, t.value


def synchronize():
    _lib.call("smm_device_sync")


class Stream:
    def __init__(self):
        h = ctypes.c_void_p()
        _lib.call("smm_stream_create", ctypes.byref(h))
        self.handle = h

    def synchronize(self):
        _lib.call("smm_stream_sync", self.handle)

    def wait_event(self, event):
        """Work queued on this stream afterwards waits for `event`."""
        _lib.call("smm_stream_wait_event", self.handle, event.handle)

    def close(self):
        if self.handle:
            _lib.call("smm_stream_destroy", self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _stream_handle(stream):
    if stream is None:
        return None
    return stream.handle if isinstance(stream, Stream) else stream


class Event:
    def __init__(self):
        h = ctypes.c_void_p()
        _lib.call("smm_event_create", ctypes.byref(h))
        self.handle = h

    def record(self, stream=None):
        _lib.call("smm_event_record", self.handle, _stream_handle(stream))

    def synchronize(self):
        _lib.call("smm_event_sync", self.handle)

    def elapsed_ms(self, stop):
        ms = ctypes.c_float(0)
        _lib.call("smm_event_elapsed_ms", self.handle, stop.handle, ctypes.byref(ms))
        return ms.value

    def __del__(self):
        try:
            if self.handle:
                _lib.call("smm_event_destroy", self.handle)
                self.handle = None
        except Exception:
            pass


class DeviceArray:
    """A C-contiguous array resident in HBM.  Owns its allocation unless it is a view.

    `layout` tags how a FIELD is laid out: "bs" (default) -- the reference's native order, horizontal
    cells fastest, (batch..., S) (regrid.py:539-541); "sb" -- batch-fastest, (S..., batch...): the batch
    values of one cell are contiguous.  `SparseOperator.apply` / `Regridder` route an "sb" field to the
    batch-fastest kernel (smm_apply_sb), whose HBM traffic equals the algorithmic bytes."""

    def __init__(self, shape, dtype, ptr=None, base=None, layout="bs"):
        self.shape = tuple(int(s) for s in np.atleast_1d(shape)) if not isinstance(shape, tuple) \
            else tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.base = base
        if layout not in ("bs", "sb"):
            raise ValueError("layout must be 'bs' (cells fastest) or 'sb' (batch fastest)")
        self.layout = layout
        if ptr is None:
            h = ctypes.c_void_p()
            _lib.call("smm_malloc", ctypes.byref(h), self.nbytes)
            self.ptr = h.value or 0
            self._owned = True
        else:
            self.ptr = int(ptr)
            self._owned = False

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def __cuda_array_interface__(self):
        """Zero-copy hand-over to array libraries that speak the protocol (``torch.as_tensor``,
        cupy, numba): the buffer stays owned by this DeviceArray."""
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (int(self.ptr), False),
                "version": 2, "strides": None}

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        shape = list(shape)
        if -1 in shape:
            i = shape.index(-1)
            known = 1
            for k, s in enumerate(shape):
                if k != i:
                    known *= s
            shape[i] = self.size // known if known else 0
        out = DeviceArray(tuple(shape), self.dtype, ptr=self.ptr, base=self.base or self, layout=self.layout)
        if out.size != self.size:
            raise ValueError(f"cannot reshape {self.shape} into {tuple(shape)}")
        return out

    def rows(self, start, stop):
        """View of rows [start, stop) along the first axis."""
        start, stop = int(start), int(stop)
        if not 0 <= start <= stop <= self.shape[0]:
            raise IndexError("row range out of bounds")
        row = self.size // self.shape[0] if self.shape[0] else 0
        return DeviceArray((stop - start,) + self.shape[1:], self.dtype,
                           ptr=self.ptr + start * row * self.dtype.itemsize, base=self.base or self,
                           layout=self.layout)

    def copy_from_host(self, host, stream=None):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.size != self.size:
            raise ValueError("size mismatch in copy_from_host")
        _lib.call("smm_memcpy_h2d", ctypes.c_void_p(self.ptr), host.ctypes.data_as(ctypes.c_void_p),
                  self.nbytes, _stream_handle(stream))
        return self

    def to_host(self, out=None, stream=None):
        if out is None:
            out = np.empty(self.shape, dtype=self.dtype)
        if out.nbytes != self.nbytes or not out.flags.c_contiguous:
            raise ValueError("to_host needs a C-contiguous buffer of the same size")
        _lib.call("smm_memcpy_d2h", out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(self.ptr),
                  self.nbytes, _stream_handle(stream))
        if stream is not None:
            _lib.call("smm_stream_sync", _stream_handle(stream))
        return out

    def fill_bytes(self, value=0, stream=None):
        _lib.call("smm_memset", ctypes.c_void_p(self.ptr), int(value), self.nbytes,
                  _stream_handle(stream))
        return self

    def fill_random(self, seed, mean=0.0, sigma=1.0, stream=None):
        """Counter-based pseudo-normal fill on the device (benchmarks, full-size tests)."""
        _lib.call("smm_fill_random", ctypes.c_void_p(self.ptr), dtype_code(self.dtype), self.size,
                  ctypes.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), float(mean), float(sigma),
                  _stream_handle(stream))
        return self

    def free(self):
        if self._owned and self.ptr:
            _lib.call("smm_free", ctypes.c_void_p(self.ptr))
        self.ptr = 0
        self._owned = False

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __repr__(self):
        return f"DeviceArray(shape={self.shape}, dtype={self.dtype}, layout={self.layout!r}, ptr=0x{self.ptr:x})"


def to_device(host, dtype=None, stream=None, layout="bs"):
    """Upload a host array.  layout="sb" tags it as a batch-fastest field (the array must already be
    ordered cells first, batch last -- e.g. a (lat, lon, time) field)."""
    host = np.asarray(host)
    if dtype is None:
        dtype = host.dtype
    return DeviceArray(host.shape, dtype, layout=layout).copy_from_host(host, stream=stream)


def aligned_pitch(n_elems, dtype, line_bytes=128):
    """Smallest row pitch (in elements) >= n_elems whose rows start on `line_bytes` boundaries: the
    tile kernels stage whole 128-B lines of a row, a row that starts mid-line costs one more line per
    staged run (config 3's 1442 x 1021 source: 14.0 vs 12.3 ms)."""
    isz = np.dtype(dtype).itemsize
    per = max(1, line_bytes // isz)
    return -(-int(n_elems) // per) * per


def to_device_pitched(host, dtype=None, stream=None):
    """Upload a host field of shape (..., S) into a DeviceArray of shape (..., pitch) whose rows start
    on 128-B lines (pitch = aligned_pitch(S)); pass it to SparseOperator.apply / OperatorGroup.apply
    as it is (they take the last axis as the row pitch)."""
    host = np.ascontiguousarray(host, dtype=dtype)
    n = host.shape[-1]
    pitch = aligned_pitch(n, host.dtype)
    out = DeviceArray(host.shape[:-1] + (pitch,), host.dtype)
    rows = host.size // n if n else 0
    isz = host.dtype.itemsize
    if rows and n:
        _lib.call("smm_memcpy2d_h2d", ctypes.c_void_p(out.ptr), pitch * isz, host.ctypes.data_as(ctypes.c_void_p),
                  n * isz, n * isz, rows, _stream_handle(stream))
    return out


def empty(shape, dtype=np.float64):
    if not isinstance(shape, tuple):
        shape = tuple(np.atleast_1d(shape).tolist())
    return DeviceArray(shape, dtype)


class _PinnedBlock:
    """Owner of one hipHostMalloc allocation (kept alive by the numpy array's base)."""

    def __init__(self, nbytes):
        h = ctypes.c_void_p()
        _lib.call("smm_host_alloc", ctypes.byref(h), max(int(nbytes), 1))
        self.ptr = h.value
        self.nbytes = int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                _lib.call("smm_host_free", ctypes.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64):
    """numpy array in page-locked host memory: smm_apply_host DMAs it without staging copies."""
    dtype = np.dtype(dtype)
    shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    n = int(np.prod(shape)) if shape else 1
    block = _PinnedBlock(n * dtype.itemsize)
    buf = (ctypes.c_char * max(block.nbytes, 1)).from_address(block.ptr)
    arr = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
    _PINNED_KEEPALIVE[id(buf)] = block
    import weakref
    weakref.finalize(buf, _PINNED_KEEPALIVE.pop, id(buf), None)
    return arr


_PINNED_KEEPALIVE = {}


class _ResultCache:
    """Page-locked buffers for the RESULTS of the host pipelines ("direct write of Y tiles to the consumer").

    `SparseOperator.apply_host` / `OperatorGroup.apply_host` return a fresh array per call, as the reference does.  A fresh
    pageable array costs its first-touch page faults inside the pipeline's copy-out and a staging copy (config 2, 512 rows:
    10 of 33 ms); a page-locked one is written by the DMA engine directly.  But hipHostMalloc is slow (it pins every page:
    ~0.17 s per GB), so it never runs on the caller's path: a call that finds no fitting block gets an ordinary array at once
    and a block of that size is prepared in the background; when the numpy array a block backs is garbage-collected the block
    returns to the cache instead of being freed.  A caller that regrids in a loop therefore gets page-locked results from its
    second or third call on (512 rows: 33.6 -> 26 ms, 2048 rows: 101 -> 80 ms); a single call is as before.  Bounded: at most
    `max_cached` bytes wait in the cache, at most `max_live` bytes are handed out at a time, small results are ordinary arrays,
    a failing hipHostMalloc just leaves the cache empty.  `SMM_RESULT_CACHE=0` switches it off."""

    def __init__(self, min_bytes=8 << 20, max_cached=4 << 30, max_live=16 << 30):
        import threading
        self.min_bytes, self.max_cached, self.max_live = int(min_bytes), int(max_cached), int(max_live)
        self.free = []            # (nbytes, _PinnedBlock), smallest first
        self.cached = 0           # bytes waiting in `free`
        self.live = 0             # bytes of blocks backing arrays that are still alive
        self.pending = {}         # nbytes -> thread preparing a block of that size
        self.lock = threading.RLock()      # re-entrant: a finaliser may fire inside a locked region of the same thread
        self.hits = self.misses = 0

    def _add(self, block):
        with self.lock:
            if self.cached + block.nbytes <= self.max_cached:
                self.free.append((block.nbytes, block))
                self.free.sort(key=lambda e: e[0])
                self.cached += block.nbytes
                return True
        return False              # the caller drops the block: its __del__ frees it

    def _release(self, key):
        block = _PINNED_KEEPALIVE.pop(key, None)
        if block is not None:
            with self.lock:
                self.live -= block.nbytes
            self._add(block)

    def _prepare(self, nbytes, device):
        try:
            if device is not None:
                _lib.call("smm_set_device", int(device))
            self._add(_PinnedBlock(nbytes))
        except Exception:          # cannot pin that much (or no device any more): ordinary arrays keep doing the job
            pass
        finally:
            with self.lock:
                self.pending.pop(nbytes, None)

    def wait(self, timeout=30.0):
        """Block until the background preparations in flight are done (tests, benchmarks of the steady state)."""
        with self.lock:
            threads = [th for th in self.pending.values() if th is not None]
        for th in threads:
            th.join(timeout)

    def empty(self, shape, dtype):
        """A C-contiguous array of `shape` / `dtype`: a recycled page-locked block if one fits, else np.empty."""
        import os
        import threading
        import weakref
        dtype = np.dtype(dtype)
        shape = tuple(int(v) for v in shape)
        n = int(np.prod(shape)) if shape else 1
        nbytes = n * dtype.itemsize
        if nbytes < self.min_bytes or os.environ.get("SMM_RESULT_CACHE") == "0":
            return np.empty(shape, dtype)
        block, spawn = None, False
        with self.lock:       # short, allocation-free where it can be: a finaliser (`_release`) may run whenever memory is allocated
            if self.live + nbytes <= self.max_live:
                for entry in self.free:
                    if nbytes <= entry[0] <= 2 * nbytes + (1 << 20):          # a fitting block, not a much larger one
                        block = entry[1]
                        break
                if block is not None:
                    self.free.remove(entry)
                    self.cached -= block.nbytes
                    self.live += block.nbytes
                    self.hits += 1
                else:
                    self.misses += 1
                    if nbytes not in self.pending and self.cached + nbytes <= self.max_cached:
                        self.pending[nbytes] = None                            # reserved; the thread is made outside the lock
                        spawn = True
        if spawn:
            try:
                dev = current_device()
            except Exception:
                dev = None
            th = threading.Thread(target=self._prepare, args=(nbytes, dev), daemon=True)
            with self.lock:
                self.pending[nbytes] = th
            th.start()
        if block is None:
            return np.empty(shape, dtype)
        buf = (ctypes.c_char * max(block.nbytes, 1)).from_address(block.ptr)
        arr = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
        _PINNED_KEEPALIVE[id(buf)] = block
        weakref.finalize(buf, self._release, id(buf))
        return arr

    def clear(self):
        self.wait()
        with self.lock:
            self.free, self.cached = [], 0


result_cache = _ResultCache()


def _join_result_cache():      # interpreter exit: no thread may still be inside hipHostMalloc when the runtime goes away
    try:
        result_cache.wait(timeout=10.0)
    except Exception:
        pass


import atexit  # noqa: E402

atexit.register(_join_result_cache)
