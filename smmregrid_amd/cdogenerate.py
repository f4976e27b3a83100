"""CdoGenerate -- same constructor / method names as the reference's
cdogenerate.py:21-23, :101-103, :345.

With a ``cdo`` binary on the box the weights come from it, exactly as in the reference:
``cdo [options] gen<method>,<target> [extra] [-sellevidx,<k>] <source> <weights file>`` with
``REMAP_EXTRAPOLATE`` / ``CDO_REMAP_NORM`` in the environment (cdogenerate.py:277-294), one run
per level for masked 3-D fields (:179-228), the result read back through `io.open_weights`.
Without it -- where this package is built and benchmarked -- the native generator (`gridgen`)
produces them for the grids and methods it knows, and says so in a WARNING: its weights are
not CDO's (a different cell geometry near coasts and poles).  Either way the hot path only
*consumes* the weights.
"""
import logging
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

from . import gridgen
from .gridmeta import CdoGrid
from .gridtype import GridType, tolist
from .xrlite import DataArray, Dataset, from_xarray, is_xarray

NATIVE_METHODS = ("con", "ycon", "con2", "bil", "bic", "nn", "dis", "laf")


class CdoGenerate:
    def __init__(self, source_grid, target_grid=None, cdo_extra=None, cdo_options=None,
                 cdo_download_path=None, cdo_icon_grids=None, cdo="cdo", loglevel='warning'):
        self.loggy = logging.getLogger("smmregrid.CdoGenerate")
        self.loggy.setLevel(getattr(logging, str(loglevel).upper(), logging.WARNING))
        self.cdo = cdo
        self.cdo_extra = tolist(cdo_extra) or []        # util.py:47-53: None -> []
        self.cdo_options = tolist(cdo_options) or []
        self.have_cdo = shutil.which(cdo) is not None
        self.env = os.environ.copy()                       # cdogenerate.py:61-66
        if cdo_download_path:
            self.env["CDO_DOWNLOAD_PATH"] = cdo_download_path
        if cdo_icon_grids:
            self.env["CDO_ICON_GRIDS"] = cdo_icon_grids
        # what the caller handed over (a path, a CDO grid name or a data object) is what `cdo` is given
        self._source_arg, self._target_arg = source_grid, target_grid
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
        parent = None
        if isinstance(obj, Dataset):
            parent = obj
            obj = next(v for v in obj.data_vars.values()
                       if GridType(v.dims).horizontal_dims and not any(t in str(v.name) for t in ("bnds", "bounds", "vertices")))
        if isinstance(obj, DataArray):
            lon = lat = None

            def degrees(coord):
                """Coordinate values in degrees: files written by CDO for unstructured grids carry radians
                (tests/data/tas-healpix2.nc of the reference: `units = "radian"`)."""
                v = np.asarray(coord.values, dtype=np.float64)
                return np.degrees(v) if str(coord.attrs.get("units", "")).lower().startswith("rad") else v

            for k in ("nav_lon", "lon", "longitude"):
                if k in obj.coords:
                    lon = degrees(obj.coords[k])
            for k in ("nav_lat", "lat", "latitude"):
                if k in obj.coords:
                    lat = degrees(obj.coords[k])
            if lon is not None and lat is not None and lon.ndim == 2 and lon.shape == lat.shape:
                # curvilinear grid (nav_lon / nav_lat style 2-D coordinates): its cell centres in storage order;
                # nn / dis work from centres, bil from the quadrilaterals of neighbouring centres, con from the cell
                # corners when the file carries them (bounds (y, x, 4))
                g = gridgen.Grid("points", lon.ravel(), lat.ravel(), name="curvilinear centres", cdo_type="curvilinear")
                g.shape2d = tuple(int(v) for v in lon.shape[::-1])          # SCRIP dims: fastest first
                g.vertices = CdoGenerate._cell_vertices(parent, obj, lon.size)
                return g
            if lon is not None and lat is not None and lon.ndim == 1 and lat.ndim == 1:
                if obj.coords[[k for k in ("lon", "longitude") if k in obj.coords][0]].dims == \
                        obj.coords[[k for k in ("lat", "latitude") if k in obj.coords][0]].dims:
                    # lon(cell), lat(cell): a list of cell centres (unstructured, HEALPix with coordinates)
                    hp = gridgen.healpix_grid_of_centers(lon, lat)
                    if hp is not None:       # HEALPix pixel centres: con / bil know the pixels, not just the centres
                        return hp
                    g = gridgen.Grid("points", lon, lat, name="cell centres", cdo_type="unstructured")
                    g.vertices = CdoGenerate._cell_vertices(parent, obj, lon.size)   # polygons, if the file has them
                    if g.vertices is None:       # a reduced (Gaussian) grid implies its cells: bands x arcs
                        g.vertices = gridgen.reduced_grid_vertices(lon, lat)
                        if g.vertices is not None:
                            g.name, g.cdo_type = "reduced grid", "gaussian_reduced"
                    return g
                def edges(coord_name, centres):
                    """nx + 1 cell edges from the file's (n, 2) bounds variable named by the coordinate's `bounds`
                    attribute (CF), when the cells are contiguous; None -> mid-points.  CDO uses a file's bounds too."""
                    if parent is None:
                        return None
                    bname = obj.coords[coord_name].attrs.get("bounds")
                    if not bname or bname not in parent:
                        return None
                    b = np.asarray(parent[bname].values, dtype=np.float64)
                    if str(parent[bname].attrs.get("units", obj.coords[coord_name].attrs.get("units", ""))).lower().startswith("rad"):
                        b = np.degrees(b)
                    if b.shape != (centres.size, 2):
                        return None
                    lo, hi = b.min(axis=1), b.max(axis=1)
                    rising = centres.size < 2 or centres[0] < centres[-1]
                    e = np.concatenate([lo[:1], hi]) if rising else np.concatenate([hi[:1], lo])
                    inner_ok = np.allclose(lo[1:], hi[:-1], atol=1e-6) if rising else np.allclose(hi[1:], lo[:-1], atol=1e-6)
                    return e if inner_ok else None
                lon_name = [k for k in ("lon", "longitude") if k in obj.coords][0]
                lat_name = [k for k in ("lat", "latitude") if k in obj.coords][0]
                try:
                    return gridgen.regular_grid_from_centers(lon, lat, lon_b=edges(lon_name, lon), lat_b=edges(lat_name, lat))
                except ValueError:
                    return gridgen.regular_grid_from_centers(lon, lat)   # unusable bounds: mid-points, either direction
        raise NotImplementedError("native weight generation supports CDO grid names "
                                  "(r<NX>x<NY>, hp<N>) and regular lon/lat data only")

    @staticmethod
    def _cell_vertices(parent, var, n_cells):
        """(lon_v, lat_v), each (cells, V) in degrees, from the bounds variables the lon / lat coordinates name (CF
        `bounds`), or None.  Unstructured meshes written by CDO pad short polygons by repeating the last vertex."""
        if parent is None:
            return None
        out = []
        for names in (("nav_lon", "longitude", "lon"), ("nav_lat", "latitude", "lat")):
            cname = next((k for k in names if k in var.coords), None)
            bname = var.coords[cname].attrs.get("bounds") if cname else None
            if not bname or bname not in parent:
                return None
            b = np.asarray(parent[bname].values, dtype=np.float64)
            units = str(parent[bname].attrs.get("units", var.coords[cname].attrs.get("units", ""))).lower()
            if units.startswith("rad"):
                b = np.degrees(b)
            b = b.reshape(-1, b.shape[-1])
            if b.shape[0] != n_cells or b.shape[1] < 3:
                return None
            out.append(b)
        return tuple(out)

    def _source_mask(self, level=None, mask_dim=None):
        """Land/sea mask from the NaNs of the source field (CDO's behaviour for data with missing values)."""
        obj = self.source_grid
        if isinstance(obj, Dataset):
            obj = next(v for v in obj.data_vars.values() if GridType(v.dims).horizontal_dims
                       and not any(t in str(v.name) for t in ("bnds", "bounds", "vertices")))
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
        if vertical_dim is not None:                 # deprecated_argument (util.py:26-39, cdogenerate.py:156)
            import warnings
            warnings.warn("vertical_dim is deprecated and will be removed in future versions. "
                          "Please use mask_dim instead.", DeprecationWarning)
            if mask_dim is None:
                mask_dim = vertical_dim
        if self.target_grid is None:
            raise TypeError('Target grid is not specified, cannot provide any regridding')
        if self.source_grid is None:
            raise TypeError('Source grid is not specified, cannot provide any regridding')
        if method not in ["bic", "bil", "con", "con2", "dis", "laf", "nn", "ycon"]:
            raise ValueError('The remap method provided is not supported!')          # cdogenerate.py:73-76
        if remap_norm not in ["fracarea", "destarea"]:
            raise ValueError('The remap normalization provided is not supported!')   # cdogenerate.py:77-78
        if self.have_cdo:
            return self._weights_with_cdo(method, extrapolate, remap_norm, mask_dim, nproc)
        # ---- no cdo binary on this box: native generator
        self.loggy.warning("no '%s' binary found: weights come from the native generator, they are not "
                           "CDO's (cell geometry differs near coasts and poles)", self.cdo)
        if method not in NATIVE_METHODS:
            raise NotImplementedError(f"method '{method}' needs the cdo binary (`cdo gen{method},<target> <source> "
                                      f"<weights>`); the native generator knows {', '.join(NATIVE_METHODS)}")
        if self.cdo_options or [e for e in self.cdo_extra if not str(e).startswith("-setgrid,")]:
            self.loggy.warning("cdo_options / cdo_extra %s %s are ignored without the cdo binary",
                               self.cdo_options, self.cdo_extra)
        # `-setgrid,<name>` among the extra CDO operators names the source grid of a file that carries
        # no coordinates (basic_test.py:15-29: healpix_0.nc + '-setgrid,hp1_nested')
        setgrid = [e.split(",", 1)[1] for e in (self.cdo_extra or []) if str(e).startswith("-setgrid,")]
        src = gridgen.parse_grid(setgrid[-1]) if setgrid else self._grid_of(self.source_grid)
        dst = self._grid_of(self.target_grid)
        if mask_dim is None:
            ds = gridgen.generate_weights(src, dst, method=method, src_mask=self._source_mask(), norm=remap_norm,
                                          extrapolate=extrapolate)
            return self._with_masked_flag(ds, None)
        obj = self.source_grid
        if isinstance(obj, Dataset):
            obj = next(v for v in obj.data_vars.values() if mask_dim in v.dims)
        levels = obj.coords[mask_dim].values
        per_level = [gridgen.generate_weights(src, dst, method=method,
                                              src_mask=self._source_mask(i, mask_dim),
                                              norm=remap_norm, extrapolate=extrapolate)
                     for i in range(len(levels))]
        ds = gridgen.stack_level_weights(per_level, levels, mask_dim=mask_dim, method=method)
        return self._with_masked_flag(ds, mask_dim)

    # ------------------------------------------------------------------ cdo subprocess path
    def _prepare_grid(self, grid, target=False):
        """cdogenerate.py:80-99: a data object is written to a temporary NetCDF file, a CDO grid
        name becomes `-const,1,<grid>` for the source and stays as it is for the target, any other
        string is a file path.  Returns (argument for the cdo command line, temp file or None)."""
        if is_xarray(grid):
            tmp = tempfile.NamedTemporaryFile(delete=False, suffix=".nc")
            tmp.close()
            grid.to_netcdf(tmp.name)
            return tmp.name, tmp.name
        if isinstance(grid, (Dataset, DataArray)):
            from .io import write_netcdf3
            tmp = tempfile.NamedTemporaryFile(delete=False, suffix=".nc")
            tmp.close()
            write_netcdf3(grid, tmp.name)
            return tmp.name, tmp.name
        if isinstance(grid, str):
            if CdoGrid(grid).grid_kind and not target:
                return f"-const,1,{grid}", None
            return grid, None
        raise TypeError('Grid must be a CDO grid string, a file path, or an xarray Dataset/DataArray.')

    def _cdo_generate_weights(self, sgrid, tgrid, method, extrapolate, remap_norm, cdo_extra_vertical=None):
        """One `cdo gen<method>` run (cdogenerate.py:234-303) -> weights Dataset."""
        from .io import open_weights
        env = dict(self.env)
        env["REMAP_EXTRAPOLATE"] = "on" if extrapolate else "off"
        env["CDO_REMAP_NORM"] = remap_norm
        with tempfile.NamedTemporaryFile(suffix=".nc") as weight_file:
            command = [self.cdo, *self.cdo_options, f"gen{method},{tgrid}",
                       *(self.cdo_extra + (tolist(cdo_extra_vertical) or [])), sgrid, weight_file.name]
            self.loggy.debug("Final CDO command: %s", command)
            try:
                subprocess.check_output(command, stderr=subprocess.STDOUT, env=env)
            except subprocess.CalledProcessError as err:
                print(err.output.decode(errors="replace"), file=sys.stderr)
                raise
            return open_weights(weight_file.name)

    def _weights_with_cdo(self, method, extrapolate, remap_norm, mask_dim, nproc):
        return self._with_masked_flag(self._cdo_weights(method, extrapolate, remap_norm, mask_dim, nproc),
                                      mask_dim)

    def _cdo_weights(self, method, extrapolate, remap_norm, mask_dim, nproc):
        """The weights as `cdo` wrote them (2-D), or stacked per level in the layout of
        cdogenerate.py:310-343 (3-D); host work only."""
        sgrid, s_tmp = self._prepare_grid(self._source_arg)
        tgrid, t_tmp = self._prepare_grid(self._target_arg, target=True)
        try:
            if not mask_dim:
                return self._cdo_generate_weights(sgrid, tgrid, method, extrapolate, remap_norm)
            src = self.source_grid
            if isinstance(src, Dataset):
                if mask_dim not in src.sizes:
                    raise KeyError(f'Cannot find vertical dim {mask_dim} in {list(src.sizes)}')
                src = next(v for v in src.data_vars.values() if mask_dim in v.dims)
            elif not isinstance(src, DataArray) or mask_dim not in src.dims:
                raise KeyError(f'Cannot find vertical dim {mask_dim} in the source grid')
            levels = src.coords[mask_dim].values if mask_dim in src.coords else np.arange(src.sizes[mask_dim])
            nvert = len(levels)
            self.loggy.info('Vertical dimension has length: %s', nvert)

            def one(lev):   # the reference forks a process per level around the same subprocess (:197-217)
                return self._cdo_generate_weights(sgrid, tgrid, method, extrapolate, remap_norm,
                                                  cdo_extra_vertical=[f"-sellevidx,{lev + 1}"])
            if nproc and nproc > 1:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=int(nproc)) as pool:
                    per_level = list(pool.map(one, range(nvert)))
            else:
                per_level = [one(lev) for lev in range(nvert)]
            return gridgen.stack_level_weights(per_level, levels, mask_dim=mask_dim, method=method)
        finally:
            for tmp in (s_tmp, t_tmp):
                if tmp and os.path.exists(tmp):
                    os.remove(tmp)

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

    def _areas_with_cdo(self, grid_arg, cdo_extra, cdo_options):
        """cdogenerate.py:362-400: `cdo [options] -f nc4 gridarea [extra] <grid> <file>`."""
        from .io import open_dataset
        sgrid, tmp = self._prepare_grid(grid_arg)
        try:
            with tempfile.NamedTemporaryFile(suffix=".nc") as areas_file:
                command = [self.cdo, *(cdo_options + ["-f", "nc4"]), "gridarea", *cdo_extra, sgrid, areas_file.name]
                self.loggy.debug("Final CDO command: %s", command)
                try:
                    subprocess.check_output(command, stderr=subprocess.STDOUT, env=self.env)
                except subprocess.CalledProcessError as err:
                    print(err.output.decode(errors="replace"), file=sys.stderr)
                    raise
                areas = open_dataset(areas_file.name)
            areas["cell_area"].attrs.update(units="m2", standard_name="area", long_name="area of grid cell")
            return areas
        finally:
            if tmp and os.path.exists(tmp):
                os.remove(tmp)

    def areas(self, target=False):
        """Cell areas in m^2 (cdogenerate.py:345-400): `cdo gridarea` when the binary exists (the source
        grid with cdo_extra / cdo_options, the target grid without, as the reference does), else computed
        here for regular and HEALPix grids and for cells that come with their vertices."""
        if target and self.target_grid is None:
            raise TypeError('Target grid is not specified, cannot provide any area')
        if self.have_cdo:
            if target:
                return self._areas_with_cdo(self._target_arg, [], [])
            return self._areas_with_cdo(self._source_arg, self.cdo_extra, self.cdo_options)
        grid = self._grid_of(self.target_grid if target else self.source_grid)
        r = 6371000.0   # CDO's PlanetRadius default
        if grid.kind != "regular":
            if getattr(grid, "cdo_type", "") == "healpix":     # equal-area pixels
                n = grid.lon.size
                return Dataset({"cell_area": (("cell",), np.full(n, 4.0 * np.pi * r * r / n), {"units": "m2"})})
            if grid.vertices is not None:                      # mesh / curvilinear cells: great-circle polygons
                area = gridgen.polygon_areas(*grid.vertices) * r * r
                if grid.shape2d is not None:
                    nx, ny = grid.shape2d
                    return Dataset({"cell_area": (("y", "x"), area.reshape(ny, nx), {"units": "m2"})})
                return Dataset({"cell_area": (("cell",), area, {"units": "m2"})})
            raise NotImplementedError("areas need a regular or HEALPix grid, or cells with their vertices (bounds "
                                      "variables in the Dataset), without `cdo`")
        area = (np.diff(np.sin(np.radians(grid.lat_b)))[:, None] *
                np.radians(np.diff(grid.lon_b))[None, :]) * r * r
        return Dataset({"cell_area": (("lat", "lon"), area, {"units": "m2"})},
                       coords={"lat": grid.lat, "lon": grid.lon})


def cdo_generate_weights(source_grid, target_grid, method="con", extrapolate=True,
                         remap_norm="fracarea", gridpath=None, icongridpath=None, cdo_extra=None,
                         cdo_options=None, mask_dim=None, vertical_dim=None, cdo="cdo", nproc=1,
                         loglevel='warning'):
    """Deprecated wrapper kept for API parity (cdogenerate.py:403-423: same keywords, same order)."""
    import warnings
    warnings.warn("cdo_generate_weights() is now deprecated, please use CdoGenerate().weights()",
                  DeprecationWarning)
    generator = CdoGenerate(source_grid=source_grid, target_grid=target_grid, loglevel=loglevel,
                            cdo_extra=cdo_extra, cdo_options=cdo_options, cdo=cdo,
                            cdo_icon_grids=icongridpath, cdo_download_path=gridpath)
    return generator.weights(method=method, extrapolate=extrapolate, remap_norm=remap_norm,
                             mask_dim=mask_dim, vertical_dim=vertical_dim, nproc=nproc)
