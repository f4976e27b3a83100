"""Sparse regrid operators resident in HBM.

`SparseOperator` stands where the reference keeps a lazy
``dask.array`` of one ``sparse.COO`` of shape (S, D) (weights.py:37-42);
`OperatorGroup` stands where it keeps the per-level Python list
(weights.py:18-23).
"""
import ctypes
import time

import numpy as np

from . import _lib
from .device import DeviceArray, dtype_code, result_cache, _stream_handle, current_device


def _cptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _launch_info(fn, handle, dt, sizes, flags):
    ints = [ctypes.c_int(0) for _ in range(5)]
    nb, lds = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.call(fn, handle, dt, *sizes, int(flags), ctypes.byref(ints[0]), ctypes.byref(ints[1]),
              ctypes.byref(ints[2]), ctypes.byref(ints[3]), ctypes.byref(nb), ctypes.byref(lds),
              ctypes.byref(ints[4]))
    return {"kernel": ("sell", "tile", "tile-dma")[ints[0].value], "j_per_block": ints[1].value,
            "rows_per_step": ints[2].value, "rows_per_block": ints[3].value, "n_blocks": nb.value,
            "lds_bytes": lds.value, "big_operator": bool(ints[4].value)}


class SparseOperator:
    """(S x D) weights matrix in HBM, built from SCRIP links (weights.py:25-44)."""

    def __init__(self, n_src, n_dst, src_address, dst_address, remap_matrix, device=None, dst_dims=None,
                 prune_zeros=False):
        """dst_dims: shape of the destination grid as the weights file gives it (`dst_grid_dims`,
        fastest dimension first); kept as metadata (save / load).
        prune_zeros: drop links whose weight is exactly zero (bilinear weights between aligned grids are
        mostly zeros); results stay bit-identical because every gathered value is finite after the fill."""
        src = np.ascontiguousarray(src_address, dtype=np.int32).ravel()
        dst = np.ascontiguousarray(dst_address, dtype=np.int32).ravel()
        w = np.asarray(remap_matrix, dtype=np.float64)
        if w.ndim == 2:
            w = w[:, 0]          # only the first weight column is used (weights.py:33)
        w = np.ascontiguousarray(w).ravel()
        if not (src.size == dst.size == w.size):
            raise ValueError("src_address, dst_address and remap_matrix differ in length")
        if device is None:
            device = current_device()
        self.device = int(device)
        h = ctypes.c_void_p()
        t0 = time.perf_counter()
        _lib.call("smm_operator_create_opt", int(n_src), int(n_dst), int(src.size), _cptr(src),
                  _cptr(dst), _cptr(w), _lib.CREATE_PRUNE_ZEROS if prune_zeros else 0, self.device, ctypes.byref(h))
        self.create_ms = (time.perf_counter() - t0) * 1e3     # sort + duplicate sum + layouts + upload
        self.dst_dims = None if dst_dims is None else tuple(int(v) for v in np.asarray(dst_dims).ravel())
        self._adopt(h)

    def _adopt(self, handle):
        self.handle = handle
        vals = [ctypes.c_int64(0) for _ in range(5)]
        _lib.call("smm_operator_info", self.handle, *[ctypes.byref(v) for v in vals])
        self.n_src, self.n_dst, self.nnz, self.n_used_src, self.max_row_nnz = [v.value for v in vals]
        self.has_imask = False
        self.has_frac = False

    @classmethod
    def from_csr(cls, n_src, n_dst, rowptr, col, val, device=None, dst_dims=None):
        """Operator from a canonical CSR (what export_csr returned): no sort, no duplicate pass.
        A non-canonical CSR (unsorted or repeated columns, bad rowptr) is a ValueError."""
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64).ravel()
        col = np.ascontiguousarray(col, dtype=np.int32).ravel()
        val = np.ascontiguousarray(val, dtype=np.float64).ravel()
        if rowptr.size != int(n_dst) + 1:
            raise ValueError(f"rowptr has {rowptr.size} entries, expected n_dst + 1 = {int(n_dst) + 1}")
        if col.size != val.size or (rowptr.size and col.size != rowptr[-1]):
            raise ValueError("col / val length differs from rowptr[-1]")
        self = cls.__new__(cls)
        self.device = int(current_device() if device is None else device)
        h = ctypes.c_void_p()
        self.dst_dims = None if dst_dims is None else tuple(int(v) for v in np.asarray(dst_dims).ravel())
        t0 = time.perf_counter()
        try:
            _lib.call("smm_operator_create_csr", int(n_src), int(n_dst), _cptr(rowptr), _cptr(col),
                      _cptr(val), self.device, ctypes.byref(h))
        except _lib.SmmError as e:
            if e.code == _lib.SMM_ERR_INVALID:
                raise ValueError(str(e)) from None
            raise
        self.create_ms = (time.perf_counter() - t0) * 1e3
        self._adopt(h)
        return self

    def save(self, path):
        """Persist the ready CSR (+ epilogue arrays if given to set_epilogue) as .npz -- the
        'native CSR cache' of SURVEY f1; reload with SparseOperator.load."""
        rowptr, col, val = self.export_csr()
        payload = {"shape": np.array([self.n_src, self.n_dst], dtype=np.int64), "rowptr": rowptr, "col": col,
                   "val": val}
        if getattr(self, "dst_dims", None):
            payload["dst_dims"] = np.asarray(self.dst_dims, dtype=np.int32)
        for name in ("_dst_imask", "_dst_frac"):
            a = getattr(self, name, None)
            if a is not None:
                payload[name[1:]] = a
        np.savez(path, **payload)

    @classmethod
    def load(cls, path, device=None):
        z = np.load(path, allow_pickle=False)
        n_src, n_dst = (int(v) for v in z["shape"])
        op = cls.from_csr(n_src, n_dst, z["rowptr"], z["col"], z["val"], device=device,
                          dst_dims=z["dst_dims"] if "dst_dims" in z.files else None)
        if "dst_imask" in z.files or "dst_frac" in z.files:
            op.set_epilogue(z["dst_imask"] if "dst_imask" in z.files else None,
                            z["dst_frac"] if "dst_frac" in z.files else None)
        return op

    # the reference's matrix is (S, D): keep .shape for code that inspects it
    @property
    def shape(self):
        return (self.n_src, self.n_dst)

    def set_epilogue(self, dst_imask=None, dst_frac=None):
        """dst_grid_imask / dst_grid_frac of the weights file (regrid.py:509-510)."""
        im = None if dst_imask is None else np.ascontiguousarray(dst_imask, dtype=np.int32).ravel()
        fr = None if dst_frac is None else np.ascontiguousarray(dst_frac, dtype=np.float64).ravel()
        for a, name in ((im, "dst_imask"), (fr, "dst_frac")):
            if a is not None and a.size != self.n_dst:
                raise ValueError(f"{name} has {a.size} entries, expected {self.n_dst}")
        _lib.call("smm_operator_set_epilogue", self.handle, _cptr(im), _cptr(fr))
        self._dst_imask, self._dst_frac = im, fr     # kept for save()
        self.has_imask = im is not None
        self.has_frac = fr is not None
        return self

    def export_csr(self):
        """(rowptr int64[D+1], col int32[nnz], val float64[nnz]) -- the canonical CSR."""
        rowptr = np.zeros(self.n_dst + 1, dtype=np.int64)
        col = np.zeros(self.nnz, dtype=np.int32)
        val = np.zeros(self.nnz, dtype=np.float64)
        _lib.call("smm_operator_export_csr", self.handle, _cptr(rowptr), _cptr(col), _cptr(val))
        return rowptr, col, val

    def plan_info(self):
        kind = ctypes.c_int(0)
        lds = ctypes.c_int64(0)
        staged = ctypes.c_int64(0)
        _lib.call("smm_operator_plan_info", self.handle, ctypes.byref(kind), ctypes.byref(lds),
                  ctypes.byref(staged))
        return {"tile_plan": bool(kind.value & 1), "tile_preferred": bool(kind.value & 2),
                "lds_bytes": lds.value, "staged_src_elems": staged.value,
                "rows_per_block": kind.value >> 8}

    def launch_info(self, n_batch, dtype=np.float64, flags=0):
        """Launch geometry `apply` would use for `n_batch` rows (nothing is launched)."""
        return _launch_info("smm_operator_launch_info", self.handle, dtype_code(np.dtype(dtype)),
                            (int(n_batch),), flags)

    def mask_apply(self, src_imask):
        """weights.py:47-52 on the device: (src_imask . W) < 0.5 ? 0 : 1."""
        src = np.ascontiguousarray(src_imask, dtype=np.int32).ravel()
        if src.size != self.n_src:
            raise ValueError(f"src_imask has {src.size} entries, expected {self.n_src}")
        out = np.empty(self.n_dst, dtype=np.int32)
        _lib.call("smm_operator_mask_apply", self.handle, _cptr(src), _cptr(out))
        return out

    def apply(self, x, y=None, masked=False, remap_area_min=0.0, out_dtype=np.float64,
              flags=0, stream=None, keep_batch_fastest=False):
        """Y = epilogue(fill(X) . W) for a device-resident X of shape (B, S), or (B, ldx) with a
        padded row pitch ldx >= S (rows that start on 128-B lines are staged without straddling).
        A field tagged batch-fastest (`x.layout == "sb"`, shape (S, B)) goes through the batch-fastest
        kernel (`apply_sb`); with keep_batch_fastest the result stays batch-fastest too, (D, B)."""
        if not isinstance(x, DeviceArray):
            raise TypeError("SparseOperator.apply takes a DeviceArray (use Regridder for host data)")
        if x.layout == "sb":
            return self.apply_sb(x, y=y, masked=masked, remap_area_min=remap_area_min, out_dtype=out_dtype,
                                 flags=flags, stream=stream, keep_batch_fastest=keep_batch_fastest)
        if keep_batch_fastest:
            raise ValueError("keep_batch_fastest needs a batch-fastest field (DeviceArray(..., layout='sb'))")
        if x.ndim != 2 or x.shape[1] < self.n_src:
            raise ValueError(f"X must be (B, >= {self.n_src}), got {x.shape}")
        n_batch = x.shape[0]
        if y is None:
            y = DeviceArray((n_batch, self.n_dst), out_dtype)
        elif y.shape != (n_batch, self.n_dst):
            raise ValueError(f"Y must be ({n_batch}, {self.n_dst}), got {y.shape}")
        fl = int(flags) | (_lib.APPLY_MASKED if masked else 0)
        _lib.call("smm_apply", self.handle, ctypes.c_void_p(x.ptr), dtype_code(x.dtype),
                  x.shape[1], ctypes.c_void_p(y.ptr), dtype_code(y.dtype), self.n_dst, n_batch,
                  float(remap_area_min), fl, _stream_handle(stream))
        return y

    def used_sources(self):
        """Ascending 0-based indices of the source cells that carry a link (length n_used_src):
        the row order of a packed batch-fastest field (apply_sb(..., packed=True))."""
        used = np.empty(self.n_used_src, dtype=np.int32)
        _lib.call("smm_operator_used_sources", self.handle, _cptr(used))
        return used

    def prepare_sb(self):
        """Upload the canonical CSR the batch-fastest kernel reads (else done by the first apply_sb)."""
        _lib.call("smm_operator_prepare_sb", self.handle)
        return self

    def apply_sb(self, x, y=None, masked=False, remap_area_min=0.0, packed=False, out_dtype=np.float64,
                 flags=0, stream=None, keep_batch_fastest=False, n_batch=None):
        """The same product for a device-resident field kept batch-fastest: x of shape (S, B) -- or
        (n_used_src, B) with packed=True, rows in `used_sources()` order -- holds the B batch values
        of each source cell contiguously.  Y is (B, D) as `apply` returns it, bit-identical to
        ``apply`` on the transposed field; HBM traffic equals the algorithmic bytes because every
        needed source cell is one contiguous run (smm_apply_sb).  keep_batch_fastest: the result
        stays batch-fastest as well -- Y (D, B), tagged layout "sb" -- which is what a following regrid
        on the target grid consumes without any transpose (SMM_APPLY_SB_Y_SB).  n_batch: batch entries
        when the last axis of x is a padded pitch (cells that start on 128-B lines -- a pitch of a multiple
        of 16 doubles -- are what the cell-staging kernel likes: every 16-entry run is then one line)."""
        if not isinstance(x, DeviceArray):
            raise TypeError("SparseOperator.apply_sb takes a DeviceArray")
        rows = self.n_used_src if packed else self.n_src
        if x.ndim != 2 or x.shape[0] != rows:
            raise ValueError(f"X must be ({rows}, B), got {x.shape}")
        ldx = x.shape[1]
        n_batch = ldx if n_batch is None else int(n_batch)
        if not 0 <= n_batch <= ldx:
            raise ValueError(f"n_batch must be within the pitch {ldx}")
        y_shape = (self.n_dst, n_batch) if keep_batch_fastest else (n_batch, self.n_dst)
        if y is None:
            y = DeviceArray(y_shape, out_dtype, layout="sb" if keep_batch_fastest else "bs")
        elif y.shape != y_shape:
            raise ValueError(f"Y must be {y_shape}, got {y.shape}")
        fl = int(flags) | (_lib.APPLY_MASKED if masked else 0) | (_lib.APPLY_SB_PACKED if packed else 0)
        if keep_batch_fastest:
            fl |= _lib.APPLY_SB_Y_SB
        _lib.call("smm_apply_sb", self.handle, ctypes.c_void_p(x.ptr), dtype_code(x.dtype), max(ldx, 1),
                  ctypes.c_void_p(y.ptr), dtype_code(y.dtype), max(y_shape[1], 1), n_batch, float(remap_area_min), fl,
                  _stream_handle(stream))
        return y

    def apply_host(self, x, out=None, masked=False, remap_area_min=0.0, out_dtype=np.float64,
                   flags=0, chunk_rows=0):
        """Same product for a host (numpy) array of shape (B, S): the rows stream through the
        library's double-buffered H2D / kernel / D2H pipeline (smm_apply_host).  Arrays from
        `pinned_empty` are DMA'd without staging copies.  Returns a (B, D) numpy array."""
        x = np.asarray(x)
        if x.dtype not in (np.float32, np.float64):
            x = x.astype(np.float64)          # result_type(x, f64), regrid.py:550
        if x.ndim != 2 or x.shape[1] != self.n_src:
            raise ValueError(f"X must be (B, {self.n_src}), got {x.shape}")
        if x.strides[1] != x.itemsize or x.strides[0] % x.itemsize or x.strides[0] < self.n_src * x.itemsize:
            x = np.ascontiguousarray(x)
        n_batch = x.shape[0]
        if out is None:
            out = result_cache.empty((n_batch, self.n_dst), out_dtype)      # page-locked and recycled when large
        if out.shape != (n_batch, self.n_dst) or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous ({n_batch}, {self.n_dst}) array")
        fl = int(flags) | (_lib.APPLY_MASKED if masked else 0)
        _lib.call("smm_apply_host", self.handle, _cptr(x), dtype_code(x.dtype),
                  x.strides[0] // x.itemsize if n_batch > 1 else max(self.n_src, 1),
                  _cptr(out), dtype_code(out.dtype), self.n_dst, n_batch, float(remap_area_min), fl,
                  int(chunk_rows))
        return out

    def close(self):
        if getattr(self, "handle", None):
            _lib.call("smm_operator_destroy", self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __repr__(self):
        return (f"SparseOperator(S={self.n_src}, D={self.n_dst}, nnz={self.nnz}, "
                f"device={self.device})")


class OperatorGroup:
    """Ordered per-level operators applied in one launch (regrid.py:387-418)."""

    def __init__(self, operators):
        self.operators = list(operators)
        if not self.operators:
            raise ValueError("OperatorGroup needs at least one operator")
        arr = (ctypes.c_void_p * len(self.operators))(*[op.handle for op in self.operators])
        h = ctypes.c_void_p()
        _lib.call("smm_group_create", arr, len(self.operators), ctypes.byref(h))
        self.handle = h
        self.n_src = self.operators[0].n_src
        self.n_dst = self.operators[0].n_dst

    def __len__(self):
        return len(self.operators)

    def plan_info(self):
        kind, spb = ctypes.c_int(0), ctypes.c_int(0)
        _lib.call("smm_group_plan_info", self.handle, ctypes.byref(kind), ctypes.byref(spb))
        return {"tile_plan": bool(kind.value & 1), "tile_preferred": bool(kind.value & 2),
                "slices_per_block": spb.value}

    def __getitem__(self, i):
        return self.operators[i]

    def _level_args(self, level_index, masked_levels, n_lev=None):
        lev = np.ascontiguousarray(level_index, dtype=np.int32).ravel()
        if n_lev is not None and lev.size != n_lev:
            raise ValueError("level_index must have one entry per data level")
        ml = None
        if masked_levels is not None:
            ml = np.ascontiguousarray(masked_levels, dtype=np.uint8).ravel()
            if ml.size != len(self.operators):
                raise ValueError("masked_levels must have one entry per group member")
        return lev, ml

    def prepare(self, level_index, masked_levels=None):
        """Upload one (level_index, masked_levels) configuration ahead of time: later `apply`
        calls with it allocate nothing and never block (smm_group_prepare)."""
        lev, ml = self._level_args(level_index, masked_levels)
        _lib.call("smm_group_prepare", self.handle, lev.size, _cptr(lev), _cptr(ml))
        return self

    def launch_info(self, n_outer, n_lev, n_inner=1, dtype=np.float64, flags=0):
        return _launch_info("smm_group_launch_info", self.handle, dtype_code(np.dtype(dtype)),
                            (int(n_outer), int(n_lev), int(n_inner)), flags)

    def apply(self, x, level_index, masked_levels=None, y=None, masked=False, remap_area_min=0.0,
              transpose=True, out_dtype=np.float64, flags=0, stream=None):
        """x: DeviceArray (n_outer, n_lev, n_inner, S) -- or (..., ldx) with a padded row pitch ldx >= S.  Returns
        (n_outer, n_inner, n_lev, D) when transpose (regrid.py:420-427) else
        (n_lev, n_outer, n_inner, D) (the concat order, regrid.py:410)."""
        if not isinstance(x, DeviceArray) or x.ndim != 4 or x.shape[3] < self.n_src:
            raise ValueError(f"X must be a DeviceArray (n_outer, n_lev, n_inner, >= {self.n_src})")
        # the last axis may be a padded row pitch (>= S): rows that start on 128-B lines (a multiple
        # of 16 doubles / 32 floats) are staged without straddling lines
        n_outer, n_lev, n_inner, S = x.shape
        D = self.n_dst
        lev, ml = self._level_args(level_index, masked_levels, n_lev)
        if transpose:
            shape = (n_outer, n_inner, n_lev, D)
            ys = (n_inner * n_lev * D, D, n_lev * D)          # (outer, lev, inner) strides
        else:
            shape = (n_lev, n_outer, n_inner, D)
            ys = (n_inner * D, n_outer * n_inner * D, D)
        if y is None:
            y = DeviceArray(shape, out_dtype)
        elif y.shape != shape:
            raise ValueError(f"Y must be {shape}, got {y.shape}")
        xs = (n_lev * n_inner * S, n_inner * S, S)
        fl = int(flags) | (_lib.APPLY_MASKED if masked else 0)
        _lib.call("smm_group_apply", self.handle, ctypes.c_void_p(x.ptr), dtype_code(x.dtype),
                  xs[0], xs[1], xs[2], ctypes.c_void_p(y.ptr), dtype_code(y.dtype),
                  ys[0], ys[1], ys[2], n_outer, n_lev, n_inner, _cptr(lev), _cptr(ml),
                  float(remap_area_min), fl, _stream_handle(stream))
        return y

    def apply_sb(self, x, level_index, masked_levels=None, y=None, masked=False, remap_area_min=0.0,
                 transpose=True, out_dtype=np.float64, flags=0, stream=None, keep_batch_fastest=False, n_batch=None):
        """Masked levels for a field kept batch-fastest per level: x is a DeviceArray (n_lev, S, B) --
        per data level the B batch values of each source cell contiguous.  Returns (B, n_lev, D) when
        transpose (regrid.py:420-427) else (n_lev, B, D); bit-identical to `apply` on the transposed field.
        keep_batch_fastest: the result stays batch-fastest per level, (n_lev, D, B) tagged "sb".
        All data levels run in one grouped launch (several for more than 88 levels), ordered on `stream`.
        n_batch: batch entries when the last axis of x is a padded pitch (see SparseOperator.apply_sb)."""
        if not isinstance(x, DeviceArray) or x.ndim != 3 or x.shape[1] != self.n_src:
            raise ValueError(f"X must be a DeviceArray (n_lev, {self.n_src}, B)")
        n_lev, S, ldx = x.shape
        B = ldx if n_batch is None else int(n_batch)
        if not 0 <= B <= ldx:
            raise ValueError(f"n_batch must be within the pitch {ldx}")
        D = self.n_dst
        lev, ml = self._level_args(level_index, masked_levels, n_lev)
        if keep_batch_fastest:
            shape, ys_lev, ys_b = (n_lev, D, B), D * max(B, 1), max(B, 1)
        else:
            shape = (B, n_lev, D) if transpose else (n_lev, B, D)
            ys_lev, ys_b = (D, n_lev * D) if transpose else (B * D, D)
        if y is None:
            y = DeviceArray(shape, out_dtype, layout="sb" if keep_batch_fastest else "bs")
        elif y.shape != shape:
            raise ValueError(f"Y must be {shape}, got {y.shape}")
        fl = int(flags) | (_lib.APPLY_MASKED if masked else 0) | (_lib.APPLY_SB_Y_SB if keep_batch_fastest else 0)
        _lib.call("smm_group_apply_sb", self.handle, ctypes.c_void_p(x.ptr), dtype_code(x.dtype), S * max(ldx, 1),
                  max(ldx, 1), ctypes.c_void_p(y.ptr), dtype_code(y.dtype), ys_lev, ys_b, B, n_lev, _cptr(lev),
                  _cptr(ml), float(remap_area_min), fl, _stream_handle(stream))
        return y

    def apply_host(self, x, level_index, masked_levels=None, masked=False, remap_area_min=0.0,
                   transpose=True, out_dtype=np.float64, flags=0, chunk_outer=0):
        """Host (numpy) variant: x of shape (n_outer, n_lev, n_inner, S); chunks of the outer
        axis stream through the group's H2D / kernel / D2H pipeline (smm_group_apply_host)."""
        x = np.asarray(x)
        if x.dtype not in (np.float32, np.float64):
            x = x.astype(np.float64)
        x = np.ascontiguousarray(x)
        if x.ndim != 4 or x.shape[3] != self.n_src:
            raise ValueError(f"X must be (n_outer, n_lev, n_inner, {self.n_src}), got {x.shape}")
        n_outer, n_lev, n_inner, _ = x.shape
        lev, ml = self._level_args(level_index, masked_levels, n_lev)
        shape = (n_outer, n_inner, n_lev, self.n_dst) if transpose else (n_lev, n_outer, n_inner, self.n_dst)
        out = result_cache.empty(shape, out_dtype)
        fl = int(flags) | (_lib.APPLY_MASKED if masked else 0)
        _lib.call("smm_group_apply_host", self.handle, _cptr(x), dtype_code(x.dtype), _cptr(out),
                  dtype_code(out.dtype), n_outer, n_lev, n_inner, int(bool(transpose)), _cptr(lev),
                  _cptr(ml), float(remap_area_min), fl, int(chunk_outer))
        return out

    def close(self):
        if getattr(self, "handle", None):
            _lib.call("smm_group_destroy", self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
