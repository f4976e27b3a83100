"""CdoGenerate -- same constructor / method names as the reference's
cdogenerate.py:21-23, :101-103, :345; the weights are produced by the native
generator (`gridgen`) because the hot path this package accelerates only
*consumes* weights and no ``cdo`` binary exists in the target environment.
Grids and methods outside the native generator raise `NotImplementedError`
naming the CDO command the reference would have run (cdogenerate.py:285-294).
"""
import logging
import os
import shutil

import numpy as np

from . import gridgen
from .gridtype import GridType, tolist
from .xrlite import DataArray, Dataset, from_xarray


class CdoGenerate:
    def __init__(self, source_grid, target_grid=None, cdo_extra=None, cdo_options=None,
                 cdo_download_path=None, cdo_icon_grids=None, cdo="cdo", loglevel='warning'):
        self.loggy = logging.getLogger("smmregrid.CdoGenerate")
        self.loggy.setLevel(getattr(logging, str(loglevel).upper(), logging.WARNING))
        self.cdo = cdo
        self.cdo_extra = tolist(cdo_extra)
        self.cdo_options = tolist(cdo_options)
        self.have_cdo = shutil.which(cdo) is not None
        self.source_grid = from_xarray(self._open_if_file(source_grid))
        self.target_grid = from_xarray(self._open_if_file(target_grid))

    @staticmethod
    def _open_if_file(grid):
        """A path to a data file stands for the grid of its fields (cdogenerate.py:184, util.py:11-31);
        other strings are CDO grid names."""
        if isinstance(grid, str) and os.path.isfile(grid):
            from .io import open_dataset
            return open_dataset(grid)
        return grid

    # ------------------------------------------------------------------ grids
    @staticmethod
    def _grid_of(obj):
        """CDO grid name, Grid, or a data object with 1-D lon/lat coordinates -> Grid."""
        if isinstance(obj, (str, gridgen.Grid)):
            return gridgen.parse_grid(obj)
        if isinstance(obj, Dataset):
            obj = next(v for v in obj.data_vars.values()
                       if GridType(v.dims).horizontal_dims)
        if isinstance(obj, DataArray):
            lon = lat = None
            for k in ("lon", "longitude"):
                if k in obj.coords:
                    lon = obj.coords[k].values
            for k in ("lat", "latitude"):
                if k in obj.coords:
                    lat = obj.coords[k].values
            if lon is not None and lat is not None and lon.ndim == 1 and lat.ndim == 1:
                if obj.coords[[k for k in ("lon", "longitude") if k in obj.coords][0]].dims == \
                        obj.coords[[k for k in ("lat", "latitude") if k in obj.coords][0]].dims:
                    # lon(cell), lat(cell): a list of cell centres (unstructured, HEALPix with coordinates)
                    return gridgen.Grid("points", lon, lat, name="cell centres", cdo_type="unstructured")
                return gridgen.regular_grid_from_centers(lon, lat)   # either latitude direction
        raise NotImplementedError("native weight generation supports CDO grid names "
                                  "(r<NX>x<NY>, hp<N>) and regular lon/lat data only")

    def _source_mask(self, level=None, mask_dim=None):
        """Land/sea mask from the NaNs of the source field (CDO's behaviour for data with missing values)."""
        obj = self.source_grid
        if isinstance(obj, Dataset):
            obj = next(v for v in obj.data_vars.values() if GridType(v.dims).horizontal_dims)
        if not isinstance(obj, DataArray):
            return None
        gt = GridType(obj.dims)
        sel = {d: 0 for d in (gt.time_dims or []) + gt.other_dims if d in obj.dims}
        if mask_dim is not None and level is not None:
            sel[mask_dim] = level
        v = obj.isel(**sel).values if sel else obj.values
        if not np.issubdtype(v.dtype, np.floating) or np.isfinite(v).all():
            return None
        return np.isfinite(v).astype(np.int32).ravel()

    # ------------------------------------------------------------------ API
    def weights(self, method="con", extrapolate=True, remap_norm="fracarea", mask_dim=None,
                vertical_dim=None, nproc=1):
        """Weights Dataset in CDO/SCRIP layout; 3-D (per level) when mask_dim is given
        (cdogenerate.py:101-228)."""
        if vertical_dim is not None and mask_dim is None:
            mask_dim = vertical_dim
        if self.target_grid is None:
            raise TypeError('Target grid is not specified, cannot provide any regridding')
        if method not in ["bic", "bil", "con", "con2", "dis", "laf", "nn", "ycon"]:
            raise KeyError(f'Unsupported method {method}')   # cdogenerate.py:73-76
        # `-setgrid,<name>` among the extra CDO operators names the source grid of a file that carries
        # no coordinates (basic_test.py:15-29: healpix_0.nc + '-setgrid,hp1_nested')
        setgrid = [e.split(",", 1)[1] for e in (self.cdo_extra or []) if str(e).startswith("-setgrid,")]
        src = gridgen.parse_grid(setgrid[-1]) if setgrid else self._grid_of(self.source_grid)
        dst = self._grid_of(self.target_grid)
        if mask_dim is None:
            ds = gridgen.generate_weights(src, dst, method=method,
                                          src_mask=self._source_mask(), norm=remap_norm)
            return self._with_masked_flag(ds, None)
        obj = self.source_grid
        if isinstance(obj, Dataset):
            obj = next(v for v in obj.data_vars.values() if mask_dim in v.dims)
        levels = obj.coords[mask_dim].values
        per_level = [gridgen.generate_weights(src, dst, method=method,
                                              src_mask=self._source_mask(i, mask_dim),
                                              norm=remap_norm)
                     for i in range(len(levels))]
        ds = gridgen.stack_level_weights(per_level, levels, mask_dim=mask_dim, method=method)
        return self._with_masked_flag(ds, mask_dim)

    @staticmethod
    def _with_masked_flag(ds, mask_dim):
        """cdogenerate.py:173-177, :221-228: pre-compute dst_grid_imask and
        `dst_grid_masked` so Regridder.__init__ can skip it (regrid.py:198-199)."""
        from .weights import (check_mask, compute_weights_matrix, compute_weights_matrix3d,
                              mask_weights)
        ops = (compute_weights_matrix3d(ds, mask_dim) if mask_dim
               else compute_weights_matrix(ds))
        ds = mask_weights(ds, ops, mask_dim)
        masked = check_mask(ds, mask_dim)
        ds["dst_grid_masked"] = DataArray(np.asarray(masked), dims=(mask_dim,) if mask_dim else ())
        for op in (ops if mask_dim else [ops]):
            op.close()
        return ds

    def areas(self, target=False):
        """Cell areas in m^2 (cdogenerate.py:345-400, `cdo gridarea`) for regular grids."""
        grid = self._grid_of(self.target_grid if target else self.source_grid)
        r = 6371000.0   # CDO's PlanetRadius default
        if grid.kind != "regular":
            if getattr(grid, "cdo_type", "") == "healpix":     # equal-area pixels
                n = grid.lon.size
                return Dataset({"cell_area": (("cell",), np.full(n, 4.0 * np.pi * r * r / n), {"units": "m2"})})
            raise NotImplementedError("areas need a regular or HEALPix grid without `cdo`")
        area = (np.diff(np.sin(np.radians(grid.lat_b)))[:, None] *
                np.radians(np.diff(grid.lon_b))[None, :]) * r * r
        return Dataset({"cell_area": (("lat", "lon"), area, {"units": "m2"})},
                       coords={"lat": grid.lat, "lon": grid.lon})


def cdo_generate_weights(source_grid, target_grid, method="con", extrapolate=True,
                         remap_norm="fracarea", vertical_dim=None, cdo_extra=None, cdo_options=None,
                         cdo="cdo", nproc=1, loglevel='warning'):
    """Deprecated wrapper kept for API parity (cdogenerate.py:403-418)."""
    import warnings
    warnings.warn("cdo_generate_weights is deprecated, use CdoGenerate().weights()",
                  DeprecationWarning)
    return CdoGenerate(source_grid, target_grid, cdo_extra=cdo_extra, cdo_options=cdo_options,
                       cdo=cdo, loglevel=loglevel).weights(method=method, extrapolate=extrapolate,
                                                           remap_norm=remap_norm,
                                                           mask_dim=vertical_dim, nproc=nproc)
