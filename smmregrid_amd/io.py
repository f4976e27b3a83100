"""Weights files without xarray/netCDF4: ``.npz`` (this package's cache format)
and NetCDF-3 classic / 64-bit-offset (``scipy.io.netcdf_file``).  HDF5-based
NetCDF-4 files need xarray (then pass the opened Dataset to Regridder)."""
import json

import numpy as np

from .xrlite import DataArray, Dataset, HAVE_XARRAY

_COORD_HINT = "__coords__"


def save_weights(weights, path):
    """Dataset -> .npz (variables, dims, attrs, coords)."""
    payload = {}
    meta = {"dims": {}, "attrs": {k: str(v) for k, v in weights.attrs.items()}, "coords": []}
    for k, v in weights.variables.items():
        payload["var__" + k] = v.values
        meta["dims"][k] = list(v.dims)
    meta["coords"] = list(weights.coords)
    payload["__meta__"] = np.array(json.dumps(meta))
    np.savez_compressed(path, **payload)


def open_weights(path):
    if str(path).endswith(".npz"):
        z = np.load(path, allow_pickle=False)
        meta = json.loads(str(z["__meta__"]))
        ds = Dataset(attrs=meta["attrs"])
        for k, dims in meta["dims"].items():
            arr = DataArray(z["var__" + k], dims=dims, name=k)
            if k in meta["coords"]:
                ds.coords[k] = arr
            else:
                ds[k] = arr
        return ds
    with open(path, "rb") as f:
        magic = f.read(4)
    if magic[:3] == b"CDF":
        from scipy.io import netcdf_file
        with netcdf_file(path, "r", mmap=False) as nc:
            ds = Dataset(attrs={k: (v.decode() if isinstance(v, bytes) else v)
                                for k, v in nc._attributes.items()})
            for k, var in nc.variables.items():
                arr = DataArray(np.array(var[...]), dims=var.dimensions, name=k)
                if var.dimensions == (k,):
                    ds.coords[k] = arr
                else:
                    ds[k] = arr
        return ds
    if HAVE_XARRAY:
        import xarray
        from .xrlite import from_xarray
        return from_xarray(xarray.open_dataset(path))
    raise OSError(f"{path}: HDF5-based NetCDF-4 cannot be read without xarray/netCDF4")
