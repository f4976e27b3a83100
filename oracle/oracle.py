"""CPU restatement of the smmregrid apply path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package ``smmregrid_amd`` never does.

PARITY UNPINNED by reference artefacts: the reference (jhardenberg/smmregrid
v0.1.6 at /root/reference) cannot be imported in this environment (ordinary
``ModuleNotFoundError`` for xarray / dask / sparse) and ships no stored weights
or golden outputs -- each of its numeric tests needs the ``cdo`` binary.  The
restatement below follows the reference's own lines plus the public semantics
of its un-vendored, unpinned dependencies (``sparse.COO``, ``dask.array``;
pyproject.toml:23-33), and is pinned by

* two independent implementations that must agree bit for bit on indices and
  to 1e-13 on values: numpy/scipy.sparse (this file) and sequential C
  (oracle.c);
* analytic known answers mirroring the reference's tests (tests/test_oracle.py);
* tests/golden/dask_statements.npz: outputs of the reference's statement sequence
  regrid.py:545-570 executed verbatim with dask.array (fill, tensordot, where's) on a
  dense copy of the weights -- only the sparse.COO constructor / summation order of the
  product remain unpinned.

Reference lines followed (paths relative to /root/reference/smmregrid):
  weights.py:25-44    compute_weights_matrix   -> coo_to_csr
  weights.py:7-23     compute_weights_matrix3d -> coo_to_csr per level, links[:link_length]
  weights.py:47-52    mask_tensordot           -> mask_apply
  weights.py:103-120  check_mask               -> check_mask
  regrid.py:536-570   apply_weights core       -> apply
  regrid.py:387-427   regrid3d loop/concat/transpose -> apply_levels
"""
import ctypes
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build_c_oracle(force=False):
    """Compile oracle.c -> liboracle.so (gcc, OpenMP)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def c_lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build_c_oracle())
        i64, p = ctypes.c_int64, ctypes.c_void_p
        lib.oracle_coo_to_csr.restype = i64
        lib.oracle_coo_to_csr.argtypes = [i64, i64, i64, p, p, p, p, p, p]
        lib.oracle_apply.restype = None
        lib.oracle_apply.argtypes = [i64, p, p, p, p, ctypes.c_int, i64, p, i64, i64,
                                     ctypes.c_int, p, p, ctypes.c_double, ctypes.c_int,
                                     ctypes.c_int]
        lib.oracle_mask_apply.restype = None
        lib.oracle_mask_apply.argtypes = [i64, p, p, p, p, p]
        lib.oracle_num_threads.restype = ctypes.c_int
        _LIB = lib
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------- operator

def coo_to_csr(n_src, n_dst, src_address, dst_address, remap_matrix):
    """weights.py:31-39 with scipy: 1-based -> 0-based, weight column 0, COO with
    summed duplicates and sorted coordinates, held as CSR (row = destination)."""
    src = np.asarray(src_address, dtype=np.int64) - 1
    dst = np.asarray(dst_address, dtype=np.int64) - 1
    w = np.asarray(remap_matrix, dtype=np.float64)
    if w.ndim == 2:
        w = w[:, 0]
    if src.size and (src.min() < 0 or src.max() >= n_src or dst.min() < 0 or dst.max() >= n_dst):
        raise ValueError("address out of range")
    m = sp.coo_matrix((w, (dst, src)), shape=(n_dst, n_src)).tocsr()
    m.sum_duplicates()
    m.sort_indices()
    return m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(np.float64)


def coo_to_csr_c(n_src, n_dst, src_address, dst_address, remap_matrix):
    """Same through oracle.c (sequential duplicate summing in link order)."""
    src = np.ascontiguousarray(src_address, dtype=np.int32)
    dst = np.ascontiguousarray(dst_address, dtype=np.int32)
    w = np.asarray(remap_matrix, dtype=np.float64)
    if w.ndim == 2:
        w = w[:, 0]
    w = np.ascontiguousarray(w)
    nnz = src.size
    rowptr = np.zeros(n_dst + 1, dtype=np.int64)
    col = np.zeros(max(nnz, 1), dtype=np.int32)
    val = np.zeros(max(nnz, 1), dtype=np.float64)
    n = c_lib().oracle_coo_to_csr(n_src, n_dst, nnz, _ptr(src), _ptr(dst), _ptr(w),
                                  _ptr(rowptr), _ptr(col), _ptr(val))
    if n < 0:
        raise ValueError("address out of range")
    return rowptr, col[:n].copy(), val[:n].copy()


# --------------------------------------------------------------------------- apply

def fill_invalid(x):
    """regrid.py:545-547: numpy.ma.fix_invalid + filled.  Non-finite entries are
    replaced by the default float fill value 1e20 cast to the array's dtype."""
    x = np.asarray(x)
    if not np.issubdtype(x.dtype, np.floating):
        return x
    bad = ~np.isfinite(x)
    if bad.any():
        x = x.copy()
        x[bad] = x.dtype.type(1e20)
    return x


def epilogue(y, masked, dst_imask, dst_frac, area_min):
    """regrid.py:553-570 in the reference's order."""
    if masked:
        y = np.where(np.asarray(dst_imask).reshape(1, -1).astype(bool), y, np.nan)
    if area_min > 0.0:
        y = np.where(np.broadcast_to(np.asarray(dst_frac), y.shape) < area_min, np.nan, y)
    with np.errstate(invalid="ignore"):
        y = np.where(y > 1e19, np.nan, y)
    return y


def apply(csr, x2d, masked=False, dst_imask=None, dst_frac=None, area_min=0.0, fill=True):
    """numpy/scipy restatement of regrid.py:545-570 on X of shape (B, S)."""
    rowptr, col, val = csr
    x2d = np.asarray(x2d)
    n_dst = rowptr.size - 1
    w = sp.csr_matrix((val, col, rowptr), shape=(n_dst, x2d.shape[1]))
    xf = fill_invalid(x2d) if fill else x2d
    y = np.asarray((w @ xf.astype(np.float64).T).T)  # tensordot(X, W(S,D), axes=1)
    return epilogue(y, masked, dst_imask, dst_frac, area_min)


def apply_c(csr, x2d, masked=False, dst_imask=None, dst_frac=None, area_min=0.0, fill=True,
            threads=1):
    """Same through oracle.c: sequential ascending-source accumulation."""
    rowptr, col, val = csr
    x2d = np.ascontiguousarray(x2d)
    if x2d.dtype not in (np.float32, np.float64):
        x2d = x2d.astype(np.float64)
    n_batch, ldx = x2d.shape
    n_dst = rowptr.size - 1
    y = np.empty((n_batch, n_dst), dtype=np.float64)
    im = None if dst_imask is None else np.ascontiguousarray(dst_imask, dtype=np.int32)
    fr = None if dst_frac is None else np.ascontiguousarray(dst_frac, dtype=np.float64)
    c_lib().oracle_apply(n_dst, _ptr(rowptr), _ptr(col), _ptr(val), _ptr(x2d),
                         0 if x2d.dtype == np.float32 else 1, ldx, _ptr(y), n_dst, n_batch,
                         int(bool(masked)), _ptr(im), _ptr(fr), float(area_min), int(bool(fill)),
                         int(threads))
    return y


def mask_apply(csr, src_imask):
    """weights.py:47-52."""
    rowptr, col, val = csr
    n_dst = rowptr.size - 1
    src = np.asarray(src_imask)
    w = sp.csr_matrix((val, col, rowptr), shape=(n_dst, src.size))
    t = w @ src.astype(np.float64)
    return np.where(t < 0.5, 0, 1).astype(np.int32)


def mask_apply_c(csr, src_imask):
    rowptr, col, val = csr
    n_dst = rowptr.size - 1
    src = np.ascontiguousarray(src_imask, dtype=np.int32)
    out = np.empty(n_dst, dtype=np.int32)
    c_lib().oracle_mask_apply(n_dst, _ptr(rowptr), _ptr(col), _ptr(val), _ptr(src), _ptr(out))
    return out


def check_mask(dst_imask):
    """weights.py:103-120: True where the destination mask is not all ones.
    (D,) -> scalar bool, (L, D) -> bool (L,)."""
    m = np.asarray(dst_imask)
    if m.ndim == 1:
        return bool(~(m == 1).all())
    return ~(m == 1).all(axis=tuple(range(1, m.ndim)))


def match_levels(weight_levels, data_levels, tol=1e-3):
    """regrid.py:386-395: nearest weights level within 1e-3, ValueError if none."""
    wl = np.asarray(weight_levels, dtype=np.float64)
    out = []
    for lev in np.asarray(data_levels, dtype=np.float64):
        i = int(np.argmin(np.abs(wl - lev)))
        if not abs(wl[i] - lev) <= tol:
            raise ValueError(f"{lev} not found in mask_dim. Available levels: {list(wl)}")
        out.append(i)
    return np.asarray(out, dtype=np.int32)


def apply_levels(csr_list, x, lev_axis, level_index, masked_levels, dst_imask, dst_frac,
                 area_min, transpose=True, use_c=True):
    """regrid.py:387-427.  x has shape (..., S) with the mask dimension at
    ``lev_axis``; per level the matching CSR / mask / frac are applied, the
    results are concatenated on a new leading level axis and (transpose=True)
    the level axis is moved to just before the horizontal axis."""
    x = np.asarray(x)
    fn = apply_c if use_c else apply
    outs = []
    for idx in range(x.shape[lev_axis]):
        widx = int(level_index[idx])
        xa = np.take(x, idx, axis=lev_axis)
        kept = xa.shape[:-1]
        y = fn(csr_list[widx], xa.reshape(-1, xa.shape[-1]), masked=bool(masked_levels[widx]),
               dst_imask=None if dst_imask is None else dst_imask[widx],
               dst_frac=None if dst_frac is None else dst_frac[widx], area_min=area_min)
        outs.append(y.reshape(kept + (y.shape[-1],)))
    out = np.stack(outs, axis=0)
    if transpose:
        out = np.moveaxis(out, 0, -2)
    return out
