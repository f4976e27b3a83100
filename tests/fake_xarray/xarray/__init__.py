"""A stand-in for the parts of xarray's object model the facade touches (tests only).

xarray is not installed in this image; the adapter code of smmregrid_amd.xrlite
(`is_xarray`, `from_xarray`, `to_xarray`; reference call sites regrid.py:251-271) would otherwise
never run.  Tests put this directory on PYTHONPATH in a subprocess so that `import xarray` finds
it.  Only attribute access the adapter performs is modelled: dims / coords / attrs / values / data
/ name on DataArray, data_vars / coords / attrs on Dataset."""
from collections import OrderedDict

import numpy as np

__version__ = "0.0-stub"


class DataArray:
    def __init__(self, data=None, dims=None, coords=None, attrs=None, name=None):
        self.data = None if data is None else (data if hasattr(data, "chunks") else np.asarray(data))
        self.dims = tuple(dims or ())
        self.attrs = dict(attrs or {})
        self.name = name
        self.coords = OrderedDict()
        for k, v in (coords or {}).items():
            if isinstance(v, DataArray):
                self.coords[k] = v
            elif isinstance(v, tuple):                       # (dims, values[, attrs])
                self.coords[k] = DataArray(v[1], dims=v[0], attrs=v[2] if len(v) > 2 else None, name=k)
            else:
                self.coords[k] = DataArray(v, dims=(k,), name=k)

    @property
    def values(self):
        return np.asarray(self.data.compute() if hasattr(self.data, "compute") else self.data)

    @property
    def shape(self):
        return () if self.data is None else tuple(self.data.shape)


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self.data_vars = OrderedDict(data_vars or {})
        self.attrs = dict(attrs or {})
        self.coords = OrderedDict()
        for k, v in (coords or {}).items():
            self.coords[k] = v if isinstance(v, DataArray) else DataArray(v, dims=(k,), name=k)
        for v in self.data_vars.values():
            for k, c in v.coords.items():
                self.coords.setdefault(k, c)

    def __getitem__(self, name):
        return self.data_vars[name]
