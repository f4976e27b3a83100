"""Weights and field files without xarray/netCDF4: ``.npz`` (this package's cache format),
NetCDF-3 classic / 64-bit-offset (``scipy.io.netcdf_file``) and HDF5-based NetCDF-4 -- through
xarray or h5py when importable, else the built-in pure-Python reader (``hdf5lite.py``)."""
import json
from collections import OrderedDict

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


def write_netcdf3(ds, path):
    """Dataset / DataArray -> NetCDF-3 (64-bit offset) with ``scipy.io.netcdf_file``: what the
    reference obtains from ``xarray.to_netcdf`` when it hands a grid to ``cdo``
    (cdogenerate.py:82-87).  NaN in floating fields is written as such with ``_FillValue = NaN``
    so that CDO sees the land/sea mask as missing values."""
    from scipy.io import netcdf_file
    if isinstance(ds, DataArray):
        da = ds
        ds = Dataset({da.name or "field": da}, coords=dict(da.coords))
    # as xarray.to_netcdf: coordinates that hang on a field but not on the Dataset are written too, and a field names
    # its auxiliary coordinates (lon / lat per cell, 2-D nav_lon / nav_lat) in the CF `coordinates` attribute -- that
    # attribute is how CDO finds the grid of an unstructured or curvilinear field
    variables = OrderedDict(ds.variables)
    aux = {}
    for name, v in ds.data_vars.items():
        for ck, c in v.coords.items():
            if c.dims and ck not in variables:
                variables[ck] = c
            if c.dims and ck not in v.dims and set(c.dims) <= set(v.dims) and (c.dims != (ck,)):
                aux.setdefault(name, []).append(ck)
    with netcdf_file(path, "w", version=2) as nc:
        for k, v in ds.attrs.items():
            setattr(nc, k, v if isinstance(v, (int, float, str)) else str(v))
        sizes = {}
        for v in variables.values():
            for d, n in zip(v.dims, v.shape):
                sizes.setdefault(d, int(n))
        for d, n in sizes.items():
            nc.createDimension(d, n)
        for k, v in variables.items():
            values = np.asarray(v.values)
            if values.dtype == np.int64:
                values = values.astype(np.int32)       # classic NetCDF has no 64-bit integers
            if values.dtype == bool:
                values = values.astype(np.int8)
            if values.dtype.kind not in "fiuS":
                continue                               # object / datetime coordinates are not grid data
            var = nc.createVariable(k, values.dtype.newbyteorder(">") if values.dtype.kind != "S" else "c",
                                    tuple(v.dims))
            var[...] = values
            for a, av in v.attrs.items():
                if isinstance(av, (int, float, str, np.generic)):
                    setattr(var, a, av)
            if k in aux and "coordinates" not in v.attrs:
                var.coordinates = " ".join(aux[k])
            if values.dtype.kind == "f" and np.isnan(values).any():
                var._FillValue = values.dtype.type(np.nan)
    return path


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
    if magic == b"GRIB":
        from .griblite import open_grib      # GRIB edition 1 (the reference reads these through cfgrib)
        return open_grib(path)
    if magic[:3] == b"CDF":
        from scipy.io import netcdf_file
        with netcdf_file(path, "r", mmap=False) as nc:
            ds = Dataset(attrs={k: (v.decode() if isinstance(v, bytes) else v)
                                for k, v in nc._attributes.items()})
            for k, var in nc.variables.items():
                attrs = {a: (v.decode() if isinstance(v, bytes) else (v.item() if getattr(v, "size", 0) == 1 else v))
                         for a, v in var._attributes.items()}
                values = np.array(var[...])
                if values.dtype.kind in "iuf" and not values.dtype.isnative:
                    values = values.astype(values.dtype.newbyteorder("="))
                values = _cf_decode(values, attrs)
                for a in ("_FillValue", "missing_value", "scale_factor", "add_offset"):
                    attrs.pop(a, None)
                arr = DataArray(values, dims=var.dimensions, name=k, attrs=attrs)
                if var.dimensions == (k,):
                    ds.coords[k] = arr
                else:
                    ds[k] = arr
        return _cf_coordinates(ds)
    if HAVE_XARRAY:
        import xarray
        from .xrlite import from_xarray
        return from_xarray(xarray.open_dataset(path))
    try:
        import h5py
    except ImportError:
        return _open_netcdf4_lite(path)      # built-in pure-Python HDF5 reader (hdf5lite.py)
    return _open_netcdf4_h5py(h5py, path)


open_dataset = open_weights                 # fields and grids come through the same readers


_NC4_INTERNAL = {"DIMENSION_LIST", "REFERENCE_LIST", "CLASS", "NAME", "_Netcdf4Dimid", "_Netcdf4Coordinates",
                 "_NCProperties", "_nc3_strict"}


def _attr_value(v):
    """HDF5 attribute -> what xarray shows: str for text, python scalar for 1-element arrays."""
    if isinstance(v, bytes):
        return v.decode("utf-8", "replace")
    if isinstance(v, np.ndarray):
        if v.dtype.kind == "S":
            return _attr_value(bytes(v.ravel()[0])) if v.size == 1 else [_attr_value(bytes(x)) for x in v.ravel()]
        if v.dtype.kind == "O":
            return v.ravel()[0] if v.size == 1 else list(v.ravel())
        if v.size == 1:
            return v.ravel()[0].item()
    return v


def _cf_decode(values, attrs):
    """xarray's default mask_and_scale for what the regridding path consumes: _FillValue /
    missing_value -> NaN, then scale_factor / add_offset.  Times stay numeric."""
    fills = [attrs[k] for k in ("_FillValue", "missing_value") if k in attrs]
    scale, offset = attrs.get("scale_factor"), attrs.get("add_offset")
    if not fills and scale is None and offset is None:
        return values
    if values.dtype.kind not in "iuf":
        return values
    out_dtype = values.dtype if values.dtype.kind == "f" else np.float64
    if values.dtype.kind in "iu" and values.dtype.itemsize <= 2 and (scale is not None or offset is not None):
        out_dtype = np.float32
    mask = np.zeros(values.shape, dtype=bool)
    for fv in fills:
        for x in np.atleast_1d(fv):
            mask |= values == x
    out = values.astype(out_dtype)
    if scale is not None:
        out = out * np.asarray(scale, dtype=out_dtype)
    if offset is not None:
        out = out + np.asarray(offset, dtype=out_dtype)
    if mask.any():
        out[mask] = np.nan
    return out


def _open_netcdf4_lite(path, decode=True):
    """NetCDF-4 (HDF5) through the built-in reader: variables with the dimension names netCDF stores
    as attached dimension scales (DIMENSION_LIST object references), coordinates, attributes."""
    from . import hdf5lite
    with hdf5lite.File(path) as f:
        nodes = {k: v for k, v in f.root.items() if isinstance(v, hdf5lite.DatasetNode)}
        by_addr = {v.addr: k for k, v in nodes.items()}
        ds = Dataset(attrs={k: _attr_value(v) for k, v in f.attrs.items() if k not in _NC4_INTERNAL})
        for name, var in nodes.items():
            at = var.attrs
            cls = at.get("CLASS")
            is_scale = cls is not None and _attr_value(cls) == "DIMENSION_SCALE"
            if is_scale and str(_attr_value(at.get("NAME", b""))).startswith("This is a netCDF dimension"):
                continue                       # a dimension without coordinate variable
            dims = []
            dimlist = at.get("DIMENSION_LIST")
            for i in range(var.ndim):
                refs = dimlist[i] if dimlist is not None and i < len(dimlist) else ()
                if len(refs) and int(refs[0]) in by_addr:
                    dims.append(by_addr[int(refs[0])])
                else:
                    dims.append(name if is_scale else f"{name}_dim{i}")
            values = var.read()
            if values is None:
                continue
            attrs = {k: _attr_value(v) for k, v in at.items() if k not in _NC4_INTERNAL}
            if decode and isinstance(values, np.ndarray):
                values = _cf_decode(values, attrs)
                for k in ("_FillValue", "missing_value", "scale_factor", "add_offset"):
                    attrs.pop(k, None)
            arr = DataArray(values, dims=dims, name=name, attrs=attrs)
            if is_scale and dims == [name]:
                ds.coords[name] = arr
            else:
                ds[name] = arr
    return _cf_coordinates(ds)


def _cf_coordinates(ds):
    """CF `coordinates` attribute (xarray's decode_coords): variables a field names as its auxiliary
    coordinates (lon / lat per cell of an unstructured or HEALPix file, 2-D nav_lon / nav_lat) become
    coordinates of the Dataset, so fields taken from it carry them."""
    names = []
    for v in ds.data_vars.values():
        for n in str(v.attrs.get("coordinates", "")).split():
            if n in ds._vars and n not in names:
                names.append(n)
    for n in names:
        ds.coords[n] = ds._vars.pop(n)
    return ds


def _open_netcdf4_h5py(h5py, path, decode=True):
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
            attrs = {k: text(v) for k, v in var.attrs.items() if k not in skip}
            values = np.asarray(var[...])
            if decode and values.dtype.kind in "fiu":
                # same CF decoding as the NetCDF-3 and built-in HDF5 readers: the field must not depend
                # on which optional package happens to be installed
                values = _cf_decode(values, attrs)
                for k in ("_FillValue", "missing_value", "scale_factor", "add_offset"):
                    attrs.pop(k, None)
            arr = DataArray(values, dims=dims, name=name, attrs=attrs)
            if is_scale and dims == [name]:
                ds.coords[name] = arr
            else:
                ds[name] = arr
    return _cf_coordinates(ds)
