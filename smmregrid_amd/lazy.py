"""Deferred results and the dask adapter of the facade.

The reference returns lazy dask arrays and keeps the kept (non-horizontal) dimensions chunked:
`dask.array.tensordot` over a field rechunked to one chunk in the horizontal dimensions
(regrid.py:29-30, :538-541, :550).  `Regridder(..., lazy=True)` reproduces that contract:

* a dask-backed field comes back dask-backed -- `map_batch_blocks` maps the GPU apply over the
  blocks of the kept dimensions with ``dask.array.map_blocks``, so nothing is launched before
  ``.compute()`` and the chunking of the kept dimensions is preserved;
* any other field comes back as a `LazyArray`: shape and dtype are known at once, the launch
  happens on the first ``.values`` / ``.compute()`` / ``numpy.asarray``.
"""
import numpy as np


class LazyArray:
    """Result of a regrid whose launch is deferred to first use."""

    def __init__(self, shape, dtype, thunk):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self._thunk = thunk
        self._value = None

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    @property
    def computed(self):
        return self._thunk is None

    def compute(self):
        """Runs the deferred apply once; the result is a numpy array (a DeviceArray result is
        copied to the host)."""
        if self._thunk is not None:
            value = self._thunk()
            value = value.to_host() if hasattr(value, "to_host") else np.asarray(value)
            self._value = value.reshape(self.shape)
            self._thunk = None
        return self._value

    values = property(compute)
    to_host = compute          # DataArray.values goes through this

    def __array__(self, dtype=None, copy=None):
        out = self.compute()
        return out if dtype is None else out.astype(dtype, copy=False)

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        new = LazyArray(np.empty(self.shape, dtype=np.bool_).reshape(shape).shape, self.dtype,
                        lambda: self.compute().reshape(shape))
        return new

    def __repr__(self):
        state = "computed" if self.computed else "deferred"
        return f"LazyArray(shape={self.shape}, dtype={self.dtype}, {state})"


def is_dask(obj):
    """True for a dask array (duck-typed: dask need not be importable)."""
    return hasattr(obj, "dask") and hasattr(obj, "chunks") and hasattr(obj, "map_blocks")


def map_batch_blocks(x, n_horizontal, tgt_shape, apply_2d, dtype=np.float64):
    """dask adapter (regrid.py:538-541, :550): `x` is a dask array whose last `n_horizontal` axes
    are the source grid.  They are rechunked to a single chunk; every block of the kept axes is
    flattened to (rows, S), handed to `apply_2d` -- the GPU apply -- and reshaped to
    (kept block..., *tgt_shape).  The result is a dask array with the kept chunking preserved."""
    kept = x.ndim - n_horizontal
    x = x.rechunk({ax: -1 for ax in range(kept, x.ndim)})
    tgt_shape = tuple(int(n) for n in tgt_shape)

    def block_fn(block):
        b = np.asarray(block)
        kept_shape = b.shape[:kept]
        rows = int(np.prod(kept_shape)) if kept_shape else 1
        y = np.asarray(apply_2d(b.reshape(rows, -1)))
        return y.reshape(kept_shape + tgt_shape)

    chunks = tuple(x.chunks[:kept]) + tuple((n,) for n in tgt_shape)
    return x.map_blocks(block_fn, dtype=np.dtype(dtype), chunks=chunks,
                        drop_axis=list(range(kept, x.ndim)),
                        new_axis=list(range(kept, kept + len(tgt_shape))))
