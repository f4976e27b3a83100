"""Grid metadata helpers with the reference's public names: `CdoGrid` (cdogrid.py:25-49),
`GridInspector` (gridinspector.py:12-112) and `GridDetector` (griddetector.py:15-132).

They are bookkeeping around the apply path -- which dimensions are horizontal / masked / time,
which variables share a grid, what kind of grid a field lives on -- and work on the lite
containers of `xrlite` (xarray objects are converted at the door).
"""
import os
import re

import numpy as np

from .gridtype import GridType
from .xrlite import DataArray, Dataset, from_xarray

# CDO grid-name grammar (the reference's table is dated CDO 2.4.4, cdogrid.py:9-22)
_CDO_GRIDS = [
    ("global_regular", r"global_\d+(\.\d+)?"),
    ("regional_regular", r"dcw:[A-Z]{2,4}(?:_\d+(\.\d+)?)?"),
    ("zonal_latitudes", r"zonal_\d+(\.\d+)?"),
    ("global_regular_NxM", r"r\d+x\d+"),
    ("one_grid_point", r"lon=(-?\d+(\.\d+)?)/lat=(-?\d+(\.\d+)?)"),
    ("gaussian_grid_F", r"F\d+"),
    ("gaussian_grid_n", r"n\d+"),
    ("icosahedral_gme", r"gme\d+"),
    ("healpix_grid", r"hp\d+(?:_(nested|ring))?"),
    ("healpix_zoom", r"hpz\d+"),
]
_CDO_GRIDS = [(kind, re.compile("^" + pat + "$")) for kind, pat in _CDO_GRIDS]


class CdoGrid:
    """A CDO grid-description string and the family it belongs to (`grid_kind` is None and
    `grid_str` 'Invalid' when the string is not a CDO grid name, e.g. a file path)."""

    def __init__(self, grid_str):
        if not isinstance(grid_str, str):
            raise TypeError("CDOGrid must be initialized with a string.")
        self.grid_kind = next((kind for kind, rx in _CDO_GRIDS if rx.match(grid_str)), None)
        self.grid_str = grid_str if self.grid_kind else "Invalid"

    def __repr__(self):
        return f"CDOGrid(grid_str='{self.grid_str}', grid_kind='{self.grid_kind}')"


_BOUNDS_MARKERS = ("bnds", "bounds", "vertices")


class GridInspector:
    """Collects the distinct `GridType`s of a DataArray / Dataset, or describes a set of CDO
    weights (`cdo_weights=True`: one gridtype whose masked dimension is the first coordinate of
    the weights, gridinspector.py:79-92)."""

    def __init__(self, data, cdo_weights=False, extra_dims=None, clean=True, loglevel="warning"):
        if isinstance(data, str):
            if not os.path.exists(data):
                raise FileNotFoundError(f"File {data} not found")
            from .io import open_weights
            data = open_weights(data)
        data = from_xarray(data)
        if not isinstance(data, (DataArray, Dataset)):
            raise TypeError("Data supplied is neither xarray Dataset or DataArray")
        self.data = data
        self.cdo_weights = cdo_weights
        self.extra_dims = extra_dims
        self.clean = clean
        self.grids = []

    @staticmethod
    def _is_bounds(name):
        return any(m in (name or "") for m in _BOUNDS_MARKERS)

    @staticmethod
    def get_gridtype_attr(gridtypes, attr):
        """Flat, duplicate-free list of one attribute over several gridtypes (gridinspector.py:242-254)."""
        out = []
        for gridtype in gridtypes:
            value = getattr(gridtype, attr, None)
            if isinstance(value, (list, tuple)):
                out.extend(value)
            elif isinstance(value, dict):
                out.extend(value.keys())
            elif isinstance(value, str):
                out.append(value)
        return list(dict.fromkeys(out))

    def get_gridtype(self):
        self.grids = []
        if self.cdo_weights:
            gridtype = GridType(dims=[], weights=self.data)
            coords = list(getattr(self.data, "coords", {}) or {})
            if coords:
                gridtype.mask_dim = coords[0]
            self.grids.append(gridtype)
            return self.grids
        arrays = self.data.data_vars.items() if isinstance(self.data, Dataset) else [(self.data.name, self.data)]
        for name, arr in arrays:
            if not isinstance(arr, DataArray):
                continue
            gt = GridType(dims=arr.dims, extra_dims=self.extra_dims)
            known = next((g for g in self.grids if g == gt), None)
            if self._is_bounds(name):
                if known is not None:
                    known.bounds.append(name)
                continue
            if known is None:
                self.grids.append(gt)
                known = gt
            known.variables[name] = {"coords": list(arr.coords)}
            known._sample = arr
        if isinstance(self.data, Dataset):      # bounds variables belong to the grid that spans their dims
            for name, arr in self.data.data_vars.items():
                if self._is_bounds(name):
                    for g in self.grids:
                        if set(g.dims) & set(arr.dims) and name not in g.bounds:
                            g.bounds.append(name)
        detector = GridDetector()
        for g in self.grids:
            g.kind = detector.detect_grid(getattr(g, "_sample", self.data))
        if self.clean:                           # grids without horizontal dims carry nothing to regrid
            self.grids = [g for g in self.grids if g.horizontal_dims]
        return self.grids


LAT_COORDS = ["lat", "latitude", "nav_lat"]
LON_COORDS = ["lon", "longitude", "nav_lon"]
_HEALPIX_DIMS = {"cell", "cells", "pix", "pixel", "healpix", "ncells"}


class GridDetector:
    """Kind of grid a field lives on: "Regular", "GaussianRegular", "GaussianReduced",
    "Curvilinear", "HEALPix", "Unstructured", "UndefinedRegular" or "Unknown"."""

    def __init__(self, lon="lon", lat="lat"):
        self.lon_coords = LON_COORDS + [lon]
        self.lat_coords = LAT_COORDS + [lat]

    @staticmethod
    def _coords(data):
        return data.coords

    @staticmethod
    def is_healpix_from_attribute(data):
        if isinstance(data, Dataset):
            if "healpix" in data.variables:
                return True
            return any(v.attrs.get("grid_mapping") == "healpix" for v in data.data_vars.values())
        return data.attrs.get("grid_mapping") == "healpix"

    @staticmethod
    def _find_healpix_dim(data):
        sizes = dict(data.sizes)
        for name, size in sizes.items():
            n = size // 12
            nside2_pow2 = size % 12 == 0 and n > 0 and (n & (n - 1)) == 0
            if nside2_pow2 and (str(name).lower() in _HEALPIX_DIMS or len(sizes) == 1):
                return name
        return None

    def detect_grid(self, data):
        data = from_xarray(data)
        if self.is_healpix_from_attribute(data) or self._find_healpix_dim(data) is not None:
            return "HEALPix"
        coords = self._coords(data)
        lat = next((c for c in self.lat_coords if c in coords), None)
        lon = next((c for c in self.lon_coords if c in coords), None)
        if not lat or not lon:
            return "Unknown"
        la, lo = coords[lat], coords[lon]
        if la.ndim == 2 and lo.ndim == 2:
            return "Curvilinear"
        if la.ndim == 1 and lo.ndim == 1:
            if la.dims != lo.dims:
                dlat, dlon = np.diff(la.values), np.diff(lo.values)
                lat_even = dlat.size == 0 or np.allclose(dlat, dlat[0])
                lon_even = dlon.size == 0 or np.allclose(dlon, dlon[0])
                if lat_even and lon_even:
                    return "Regular"
                if lon_even:
                    return "GaussianRegular"
                return "UndefinedRegular"
            south = la.values[la.values < 0]
            _, counts = np.unique(south, return_counts=True)
            if counts.size > 1 and np.all(np.diff(counts) > 0):
                return "GaussianReduced"       # more points per latitude circle towards the equator
            return "Unstructured"
        return "Unknown"
