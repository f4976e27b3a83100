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
    try:
        import h5py
    except ImportError:
        raise OSError(f"{path}: HDF5-based NetCDF-4 needs xarray, netCDF4 or h5py")
    return _open_netcdf4_h5py(h5py, path)


def _open_netcdf4_h5py(h5py, path):
    """NetCDF-4 (HDF5) weights through h5py: variables, their dimension names (from the
    dimension scales netCDF attaches), global and variable attributes."""
    def text(v):
        if isinstance(v, bytes):
            return v.decode()
        if isinstance(v, np.ndarray) and v.dtype.kind in "SO" and v.size == 1:
            return text(v.ravel()[0])
        return v

    skip = {"DIMENSION_LIST", "REFERENCE_LIST", "CLASS", "NAME", "_Netcdf4Dimid", "_Netcdf4Coordinates",
            "_NCProperties"}
    with h5py.File(path, "r") as f:
        ds = Dataset(attrs={k: text(v) for k, v in f.attrs.items() if k not in skip})
        for name, var in f.items():
            if not isinstance(var, h5py.Dataset):
                continue
            is_scale = var.attrs.get("CLASS", b"") == b"DIMENSION_SCALE"
            if is_scale and str(text(var.attrs.get("NAME", b""))).startswith("This is a netCDF dimension"):
                continue                       # a dimension without coordinate variable
            dims = []
            for i in range(var.ndim):
                has_scale = len(var.dims[i]) > 0
                dims.append(var.dims[i][0].name.split("/")[-1] if has_scale
                            else (name if is_scale else f"{name}_dim{i}"))
            arr = DataArray(np.asarray(var[...]), dims=dims, name=name,
                            attrs={k: text(v) for k, v in var.attrs.items() if k not in skip})
            if is_scale and dims == [name]:
                ds.coords[name] = arr
            else:
                ds[name] = arr
    return ds
