"""Minimal labelled containers used when xarray is not importable.

The reference speaks xarray everywhere (regrid.py:251-271).  This package runs
in environments without xarray, so the facade works on two tiny stand-ins with
the handful of attributes the hot path reads (`dims`, `sizes`, `attrs`,
`coords`, `values`, `name`, `isel`).  Real xarray objects are converted with
`from_xarray` / `to_xarray` at the facade boundary.
"""
from collections import OrderedDict

import numpy as np

try:  # pragma: no cover - depends on the environment
    import xarray as _xarray
except Exception:  # ModuleNotFoundError in this image
    _xarray = None

HAVE_XARRAY = _xarray is not None


class DataArray:
    """ndarray (or DeviceArray) + dimension names + coords + attrs."""

    def __init__(self, data=None, dims=None, coords=None, attrs=None, name=None):
        self.data = data
        if data is None:
            self.dims = ()
        else:
            shape = tuple(data.shape)
            if dims is None:
                dims = tuple(f"dim_{i}" for i in range(len(shape)))
            self.dims = tuple(dims)
            if len(self.dims) != len(shape):
                raise ValueError(f"{len(self.dims)} dims for a {len(shape)}-d array")
        self.coords = OrderedDict()
        for k, v in (coords or {}).items():
            self.coords[k] = v if isinstance(v, DataArray) else DataArray(np.asarray(v), dims=(k,) if np.ndim(v) == 1 else ())
        self.attrs = dict(attrs or {})
        self.name = name

    @property
    def shape(self):
        return () if self.data is None else tuple(self.data.shape)

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def dtype(self):
        return None if self.data is None else self.data.dtype

    @property
    def sizes(self):
        return OrderedDict(zip(self.dims, self.shape))

    @property
    def values(self):
        d = self.data
        return d.to_host() if hasattr(d, "to_host") else np.asarray(d)

    def isel(self, **indexers):
        data = self.values
        dims = list(self.dims)
        index = [slice(None)] * len(dims)
        drop = []
        for dim, idx in indexers.items():
            ax = dims.index(dim)
            index[ax] = idx
            if np.ndim(idx) == 0 and not isinstance(idx, slice):
                drop.append(dim)
        out = data[tuple(index)]
        new_dims = [d for d in dims if d not in drop]
        coords = OrderedDict()
        for k, c in self.coords.items():
            if set(c.dims) & set(indexers):
                sub = {d: indexers[d] for d in c.dims if d in indexers}
                coords[k] = c.isel(**sub)
            else:
                coords[k] = c
        return DataArray(out, dims=new_dims, coords=coords, attrs=self.attrs, name=self.name)

    def __repr__(self):
        return f"<smmregrid_amd.DataArray {self.name!r} {dict(self.sizes)} {self.dtype}>"


class Dataset:
    """Ordered name -> DataArray mapping with shared attrs (weights files, multi-variable input)."""

    def __init__(self, data_vars=None, coords=None, attrs=None):
        self._vars = OrderedDict()
        self.coords = OrderedDict()
        for k, v in (coords or {}).items():
            self.coords[k] = v if isinstance(v, DataArray) else DataArray(np.asarray(v), dims=(k,), name=k)
        for k, v in (data_vars or {}).items():
            self[k] = v
        self.attrs = dict(attrs or {})

    def __setitem__(self, name, value):
        if isinstance(value, tuple):
            dims, data = value[0], np.asarray(value[1])
            attrs = value[2] if len(value) > 2 else None
            value = DataArray(data, dims=dims, attrs=attrs)
        if value.name is None:
            value.name = name
        self._vars[name] = value

    def __getitem__(self, name):
        if name in self._vars:
            # as in xarray, a variable taken from a Dataset carries the coordinates on its dimensions
            var = self._vars[name]
            extra = {k: c for k, c in self.coords.items()
                     if k not in var.coords and c.dims and set(c.dims) <= set(var.dims)}
            if not extra:
                return var
            coords = OrderedDict(var.coords)
            coords.update(extra)
            out = DataArray(var.data, dims=var.dims, coords=coords, name=var.name)
            out.attrs = var.attrs          # shared, as in xarray: ds['v'].attrs[...] = x changes the Dataset
            return out
        if name in self.coords:
            return self.coords[name]
        raise KeyError(name)

    def __contains__(self, name):
        return name in self._vars or name in self.coords

    @property
    def data_vars(self):
        return OrderedDict((k, self[k]) for k in self._vars)

    def __getattr__(self, name):
        try:
            return self.__getitem__(name)
        except KeyError:
            raise AttributeError(name)

    @property
    def variables(self):
        out = OrderedDict(self.coords)
        out.update(self.data_vars)
        return out

    @property
    def sizes(self):
        out = OrderedDict()
        for v in self.variables.values():
            for d, n in v.sizes.items():
                out.setdefault(d, n)
        return out

    @property
    def dims(self):
        return self.sizes

    def map(self, func, keep_attrs=True):
        # as xarray.Dataset.map: the new Dataset is built from the results, so its coordinates are
        # theirs (target lat/lon after a regrid), not this Dataset's
        out = Dataset(attrs=self.attrs if keep_attrs else None)
        for k, v in self.data_vars.items():
            res = func(v)
            out[k] = res
            for ck, cv in res.coords.items():
                if ck not in out.coords and cv.dims:
                    out.coords[ck] = cv
        return out

    def drop_vars(self, names):
        out = Dataset(attrs=self.attrs, coords=self.coords)
        for k, v in self.data_vars.items():
            if k not in names:
                out[k] = v
        return out

    def __repr__(self):
        return f"<smmregrid_amd.Dataset vars={list(self.data_vars)} sizes={dict(self.sizes)}>"


def is_xarray(obj):
    return HAVE_XARRAY and isinstance(obj, (_xarray.DataArray, _xarray.Dataset))


def from_xarray(obj):
    """xarray.DataArray / Dataset -> lite containers (values are loaded)."""
    if not is_xarray(obj):
        return obj
    if isinstance(obj, _xarray.DataArray):
        coords = OrderedDict()
        for k, c in obj.coords.items():
            coords[k] = DataArray(np.asarray(c.values), dims=c.dims, attrs=dict(c.attrs), name=k)
        # a dask-backed field keeps its dask array (Regridder(lazy=True) maps the apply over its blocks;
        # the eager path computes it when it reads the field)
        raw = obj.data
        keep = isinstance(raw, np.ndarray) or (hasattr(raw, "dask") and hasattr(raw, "chunks"))
        return DataArray(raw if keep else np.asarray(obj.values),
                         dims=obj.dims, coords=coords, attrs=dict(obj.attrs), name=obj.name)
    ds = Dataset(attrs=dict(obj.attrs))
    for k, c in obj.coords.items():
        ds.coords[k] = DataArray(np.asarray(c.values), dims=c.dims, attrs=dict(c.attrs), name=k)
    for k, v in obj.data_vars.items():
        ds[k] = from_xarray(v)
    return ds


def _lazy_payload(data):
    """What an xarray.DataArray should wrap: dask arrays stay dask; a deferred `LazyArray` becomes a
    one-chunk dask array when dask is importable (the launch still waits for .compute()), else it is
    computed; DeviceArray / numpy go through as host values."""
    if hasattr(data, "dask") and hasattr(data, "chunks"):
        return data
    from .lazy import LazyArray
    if isinstance(data, LazyArray) and not data.computed:
        try:
            import dask
            import dask.array as da
            return da.from_delayed(dask.delayed(data.compute)(), shape=data.shape, dtype=data.dtype)
        except ImportError:
            pass
    return data.to_host() if hasattr(data, "to_host") else np.asarray(data)


def to_xarray(obj):
    """lite containers -> xarray (only when xarray is importable)."""
    if not HAVE_XARRAY:
        return obj
    if isinstance(obj, DataArray):
        if obj.data is None:
            return _xarray.DataArray(data=None)
        coords = {k: (c.dims, c.values, c.attrs) for k, c in obj.coords.items()}
        return _xarray.DataArray(_lazy_payload(obj.data), dims=obj.dims, coords=coords, attrs=obj.attrs,
                                 name=obj.name)
    if isinstance(obj, Dataset):
        return _xarray.Dataset({k: to_xarray(v) for k, v in obj.data_vars.items()},
                               attrs=obj.attrs)
    return obj
