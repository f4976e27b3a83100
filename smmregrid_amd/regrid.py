"""Regridder facade -- drop-in for smmregrid.regrid.Regridder on the apply path.

Same constructor keywords, methods and exceptions as the reference
(regrid.py:55-59, :233, :273, :339, :429, :458, :656); the arithmetic that the
reference hands to dask/sparse (regrid.py:545-570) runs in the HIP library.
Inputs may be xarray objects (when xarray is importable) or the lite
containers of `smmregrid_amd.xrlite`; the result has the type of the input.
A field whose ``data`` is a `DeviceArray` stays in HBM (no host copies).

By default the result is eager (numpy- or HBM-backed).  ``Regridder(..., lazy=True)`` gives the
reference's contract (regrid.py:29-30): nothing is launched before ``.values`` / ``.compute()``;
a dask-backed field comes back dask-backed with the chunking of its kept dimensions preserved
(`lazy.map_batch_blocks`), any other field as a `lazy.LazyArray`.
"""
import logging
import math
import os

import numpy as np

from .device import DeviceArray
from .gridtype import GridType, tolist
from .lazy import LazyArray, is_dask, map_batch_blocks
from .operator import OperatorGroup
from .weights import (_level_slice, check_mask, compute_weights_matrix, compute_weights_matrix3d,
                      mask_weights)
from .xrlite import DataArray, Dataset, from_xarray, is_xarray, to_xarray

DEFAULT_AREA_MIN = 0.5  # default minimum area for conservative remapping (regrid.py:49)


def _logger(level):
    log = logging.getLogger("smmregrid.Regrid")
    log.setLevel(getattr(logging, str(level).upper(), logging.WARNING))
    return log


def _remove_degenerate_axes(a):
    """dimension.py:22-37 of the reference: collapse axes along which a 2-D
    coordinate array does not vary."""
    a = np.asarray(a)
    if a.ndim != 2:
        return a
    if (a == a[0:1, :]).all():
        return a[0, :]
    if (a == a[:, 0:1]).all():
        return a[:, 0]
    return a


class Regridder(object):
    """Main regridding class (reference: regrid.py:52)."""

    def __init__(self, source_grid=None, target_grid=None, weights=None,
                 method='con', remap_area_min=DEFAULT_AREA_MIN, transpose=True, mask_dim=None,
                 vertical_dim=None, horizontal_dims=None, cdo_extra=None, cdo_options=None,
                 check_nan=False, cdo='cdo', loglevel='WARNING', device=None, out_dtype=np.float64,
                 lazy=False, prune_zero_weights=False, keep_batch_fastest=False):
        if (source_grid is None or target_grid is None) and (weights is None):
            raise ValueError("Either weights or source_grid/target_grid must be supplied")

        if vertical_dim is not None:  # deprecated_argument (util.py:23-40)
            import warnings
            warnings.warn("'vertical_dim' is deprecated, use 'mask_dim'", DeprecationWarning)
            if mask_dim is None:
                mask_dim = vertical_dim

        self.loggy = _logger(loglevel)
        self.loglevel = loglevel
        self.transpose = transpose
        self.device = device
        self.lazy = bool(lazy)
        # opt-in: links of weight exactly 0 (3 of 4 bilinear links between aligned grids) are dropped when
        # the operators are built; results are bit-identical (SMM_CREATE_PRUNE_ZEROS)
        self.prune_zero_weights = bool(prune_zero_weights)
        # Device-resident fields kept batch-fastest (`DeviceArray(..., layout="sb")`: horizontal
        # dimensions first, e.g. (lat, lon, time)) run through the batch-fastest kernel; with
        # keep_batch_fastest the result stays in that layout too -- (lat, lon, time) on the target grid,
        # HBM-resident -- so a second Regridder consumes it without a transpose.
        self.keep_batch_fastest = bool(keep_batch_fastest)
        # the reference always yields float64 (result_type(x, f64)); float32 is an opt-in narrowing store
        self.out_dtype = np.dtype(out_dtype)
        if self.out_dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError('out_dtype must be float32 or float64')
        mask_dim = tolist(mask_dim)
        horizontal_dims = tolist(horizontal_dims)
        self.extra_dims = {'mask': mask_dim, 'horizontal': horizontal_dims}

        self.remap_area_min = float(remap_area_min)
        if self.remap_area_min < 0.0 or self.remap_area_min > 1.0:
            raise ValueError('The remap_area_min provided must be between 0.0 and 1.0')

        if weights is not None:
            self.init_mode = 'weights'
            self.grids = self._gridtype_from_weights(weights)
        else:
            self.init_mode = 'grids'
            from .cdogenerate import CdoGenerate
            if isinstance(source_grid, str) and not os.path.isfile(source_grid):
                raise FileNotFoundError(f'Cannot find grid file {source_grid}')      # regrid.py:133-138
            if isinstance(source_grid, str) and os.path.isfile(source_grid):
                from .io import open_dataset       # regrid.py:133-136: a data file as source grid
                source_grid = open_dataset(source_grid)
            source_grid_array = from_xarray(source_grid)
            self.grids = self._gridtype_from_data(source_grid_array)
            if len(self.grids) == 0:
                raise ValueError('Cannot find any gridtype in your data, aborting!')
            for index, gridtype in enumerate(self.grids):
                if check_nan:
                    # NaN pattern varying along a non-time dimension -> that dimension is masked
                    gridtype = self._check_nan_variation(source_grid_array, gridtype)
                    self.grids[index] = gridtype
                # weights are generated from one variable of this gridtype (regrid.py:171-177)
                sample = next(iter(gridtype.variables.values()), source_grid_array)
                if isinstance(source_grid_array, Dataset) and isinstance(sample, DataArray):
                    # ... together with the grid's bounds variables, as the reference stores them (regrid.py:173-175;
                    # a file name goes to cdo whole, :168-169): the native generator takes a lon/lat grid's cell edges
                    # and a mesh's cell polygons from them, as CDO does.  Bounds are the variables the reference's name
                    # rule finds plus those the horizontal coordinates name in their CF `bounds` attribute
                    # (`bounds_nav_lon` of a CMOR ocean file does not end in `_bounds`)
                    named = [c.attrs.get("bounds") for k, c in sample.coords.items() if "time" not in str(k)]
                    wanted = [b for b in dict.fromkeys(list(gridtype.bounds) + named)
                              if b and b in source_grid_array and "time" not in b]
                    if wanted:
                        picked = Dataset({sample.name: sample}, attrs=source_grid_array.attrs)
                        for bname in wanted:
                            picked[bname] = source_grid_array[bname]
                        sample = picked
                generator = CdoGenerate(sample, target_grid, cdo=cdo,
                                        cdo_options=cdo_options, cdo_extra=cdo_extra,
                                        loglevel=loglevel)
                gridtype.weights = generator.weights(method=method, mask_dim=gridtype.mask_dim)

        for gridtype in self.grids:
            self._setup_operators(gridtype)

    # ------------------------------------------------------------------ init helpers
    def _setup_operators(self, gridtype):
        """regrid.py:189-203: operators, destination mask, `masked` flags; then the
        epilogue vectors are placed next to the operators in HBM."""
        w = gridtype.weights
        if gridtype.mask_dim:
            gridtype.weights_matrix = compute_weights_matrix3d(w, gridtype.mask_dim, device=self.device,
                                                               prune_zeros=self.prune_zero_weights)
        else:
            gridtype.weights_matrix = compute_weights_matrix(w, device=self.device,
                                                             prune_zeros=self.prune_zero_weights)

        if "dst_grid_masked" in w.variables:
            gridtype.masked = np.asarray(w["dst_grid_masked"].values)
            if gridtype.masked.ndim == 0:
                gridtype.masked = bool(gridtype.masked)
        else:
            gridtype.weights = mask_weights(w, gridtype.weights_matrix, gridtype.mask_dim)
            gridtype.masked = check_mask(gridtype.weights, gridtype.mask_dim)
        w = gridtype.weights

        if gridtype.mask_dim:
            for i, op in enumerate(gridtype.weights_matrix):
                op.set_epilogue(_level_slice(w["dst_grid_imask"], gridtype.mask_dim, i),
                                _level_slice(w["dst_grid_frac"], gridtype.mask_dim, i)
                                if "dst_grid_frac" in w else None)
            gridtype.group = OperatorGroup(gridtype.weights_matrix)
        else:
            gridtype.weights_matrix.set_epilogue(
                w["dst_grid_imask"].values,
                w["dst_grid_frac"].values if "dst_grid_frac" in w else None)

    def _gridtype_from_weights(self, weights):
        """regrid.py:205-223 + gridinspector.py:79-92."""
        if isinstance(weights, str):
            from .io import open_weights
            weights = open_weights(weights)
        weights = from_xarray(weights)
        if not isinstance(weights, Dataset):
            raise TypeError('weights must be a Dataset or a path to a weights file')
        gridtype = GridType(dims=[], weights=weights)
        if weights.coords:
            gridtype.mask_dim = list(weights.coords)[0]
        self.extra_dims['mask'] = [gridtype.mask_dim]
        return [gridtype]

    def _check_nan_variation(self, source_grid, gridtype):
        """regrid.py:630-653 + util.py:56-85: if the NaN mask of the first variable (first time
        step) changes along one of the unclassified dimensions, treat it as the masked dimension."""
        if gridtype.mask_dim:
            return gridtype
        arrays = list(source_grid.data_vars.values()) if isinstance(source_grid, Dataset) else [source_grid]
        arr = next((a for a in arrays if isinstance(a, DataArray)
                    and GridType(a.dims, extra_dims=self.extra_dims) == gridtype), None)
        if arr is None or not gridtype.other_dims:
            return gridtype
        if gridtype.time_dims and gridtype.time_dims[0] in arr.dims:
            arr = arr.isel(**{gridtype.time_dims[0]: 0})
        vals = arr.values
        if not np.issubdtype(vals.dtype, np.floating):
            return gridtype
        nan_mask = np.isnan(vals)
        nan_dims = [d for d in gridtype.other_dims if d in arr.dims
                    and np.diff(nan_mask.astype(np.int8), axis=arr.dims.index(d)).astype(bool).any()]
        if not nan_dims:
            return gridtype
        self.loggy.warning('Found NaN variation in dimensions: %s', nan_dims)
        self.extra_dims['mask'] = nan_dims
        new = GridType(dims=gridtype.dims + gridtype.other_dims + (gridtype.time_dims or []),
                       extra_dims=self.extra_dims)
        new.variables = gridtype.variables
        return new

    def _gridtype_from_data(self, data):
        grids = []
        arrays = data.data_vars.values() if isinstance(data, Dataset) else [data]
        for arr in arrays:
            if not isinstance(arr, DataArray):
                continue
            name = arr.name or ''
            if any(s in name for s in ("bnds", "bounds", "vertices")):
                continue
            gt = GridType(dims=arr.dims, extra_dims=self.extra_dims)
            if not gt.horizontal_dims:
                continue
            known = next((g for g in grids if g == gt), None)
            if known is None:
                grids.append(gt)
                known = gt
            known.variables[name] = arr          # gridinspector.py: variables living on this grid
        if isinstance(data, Dataset):            # gridinspector.py:183-221: spatial bounds variables go with the grid
            for name, arr in data.data_vars.items():
                spatial = (name.endswith("_bnds") or name.endswith("_bounds") or name == "vertices") and "time" not in name
                for gt in grids:
                    if spatial and set(gt.dims) & set(arr.dims) and name not in gt.bounds:
                        gt.bounds.append(name)
        return grids

    # ------------------------------------------------------------------ public API
    def regrid(self, source_data):
        """regrid.py:233-271."""
        was_xarray = is_xarray(source_data)
        data = from_xarray(source_data)
        if isinstance(data, Dataset):
            datagrids = self._gridtype_from_data(data)
            if len(datagrids) > 1 and self.init_mode == 'weights':
                raise ValueError(
                    f'Cannot process data with {len(datagrids)} GridType initializing from weights')
            out = data.map(self.regrid_array, keep_attrs=True)
            degen = [k for k, v in out.data_vars.items() if v.dims == ()]
            out = out.drop_vars(degen)
            return to_xarray(out) if was_xarray else out
        if isinstance(data, DataArray):
            out = self.regrid_array(data)
            return to_xarray(out) if was_xarray else out
        raise TypeError('The object provided is not a Xarray object!')

    def regrid_array(self, source_data):
        """regrid.py:273-312."""
        source_data = from_xarray(source_data)
        scalars = [name for name, coord in source_data.coords.items() if coord.dims == ()]
        if scalars:                                   # regrid.py:288-294
            self.loggy.warning("Found scalar coordinates %s. If have selected a along a masked dimensions,"
                               "regridding might fail. Please consider subsetting with [] or with slice", scalars)
        datagridtype = GridType(dims=source_data.dims, extra_dims=self.extra_dims)
        name = source_data.name or ''
        is_bounds = (name.endswith('_bnds') or name.endswith('_bounds') or name == 'vertices') and 'time' not in name
        if not (datagridtype.horizontal_dims or datagridtype.mask_dim) or is_bounds:
            # GridInspector finds no grid in such a variable (spatial bounds are skipped, gridinspector.py:70-78; a grid
            # without horizontal or mask dimension is cleaned away, :133-143 -- time_bnds(time, bnds) is one): nothing
            # to regrid, and the empty result is dropped from a Dataset (regrid.py:308-312, :262-264)
            return DataArray(data=None)
        if datagridtype.mask_dim:
            return self.regrid3d(source_data, datagridtype)
        return self.regrid2d(source_data, datagridtype)

    def _get_gridtype(self, datagridtype):
        """regrid.py:324-337."""
        if self.init_mode == 'weights':
            self.grids[0].dims = datagridtype.dims
            self.grids[0].horizontal_dims = datagridtype.horizontal_dims
            self.grids[0].other_dims = datagridtype.other_dims
        return next((grid for grid in self.grids if grid == datagridtype), None)

    def regrid2d(self, source_data, datagridtype):
        """regrid.py:429-456."""
        gridtype = self._get_gridtype(datagridtype)
        if gridtype is None:
            return DataArray(data=None)
        return self.apply_weights(source_data, gridtype.weights,
                                  weights_matrix=gridtype.weights_matrix,
                                  masked=gridtype.masked,
                                  horizontal_dims=gridtype.horizontal_dims)

    # ------------------------------------------------------------------ apply (2-D)
    def _target_layout(self, weights):
        """regrid.py:572-579: target shape/dims from dst_grid_dims."""
        dst_grid_shape = np.asarray(weights["dst_grid_dims"].values).astype(np.int64)
        rank = dst_grid_shape.size
        if rank == 2:
            return [int(dst_grid_shape[1]), int(dst_grid_shape[0])], ["i", "j"]
        if rank == 1:
            return [int(dst_grid_shape[0])], ['cell']
        raise ValueError('Unknown dimensional target grid')

    def _finish(self, data, dims, source_data, kept_dims, weights, tgt_shape, tgt_dims):
        """regrid.py:586-626: coordinates, lat/lon in degrees, attrs."""
        coords = {k: v for k, v in source_data.coords.items() if set(v.dims).issubset(kept_dims)}
        axis_scale = 180.0 / math.pi
        lat = _remove_degenerate_axes(
            np.asarray(weights["dst_grid_center_lat"].values).reshape(tgt_shape))
        lon = _remove_degenerate_axes(
            np.asarray(weights["dst_grid_center_lon"].values).reshape(tgt_shape))
        lat = np.round(lat * axis_scale, 10)
        lon = np.round(lon * axis_scale, 10)
        dims = list(dims)
        if tgt_dims == ["i", "j"] and lat.ndim == 1 and lon.ndim == 1:
            dims = [{"i": "lat", "j": "lon"}.get(d, d) for d in dims]
            lat_dims, lon_dims = ("lat",), ("lon",)
        else:
            lat_dims = lon_dims = tuple(tgt_dims) if lat.ndim == len(tgt_dims) else (tgt_dims[0],)
        out = DataArray(data, dims=dims, coords=coords, attrs=dict(source_data.attrs),
                        name=source_data.name)
        out.coords["lat"] = DataArray(lat, dims=lat_dims, name="lat", attrs={
            "units": "degrees_north", "standard_name": "latitude", "axis": "Y"})
        out.coords["lon"] = DataArray(lon, dims=lon_dims, name="lon", attrs={
            "units": "degrees_east", "standard_name": "longitude", "axis": "X"})
        out.attrs.pop('CDI_grid_type', None)
        return out

    def apply_weights(self, source_data, weights, weights_matrix=None, masked=True,
                      horizontal_dims=None):
        """regrid.py:458-628 for one 2-D operator."""
        source_data = from_xarray(source_data)
        weights = from_xarray(weights)
        name = source_data.name or ''
        if any(s in name for s in ("bnds", "bounds", "vertices")):
            if 'time' in name:
                return source_data
            return DataArray(data=None)

        if not any(x in source_data.dims for x in (horizontal_dims or [])):
            self.loggy.error("None of dimensions on which we can interpolate is found in the DataArray.")
            raise KeyError('Dimensions mismatch')

        kept_dims = [d for d in source_data.dims if d not in horizontal_dims]
        kept_shape = [source_data.sizes[d] for d in kept_dims]
        tgt_shape, tgt_dims = self._target_layout(weights)

        if weights_matrix is None:
            weights_matrix = compute_weights_matrix(weights, device=self.device)
            weights_matrix.set_epilogue(
                weights["dst_grid_imask"].values,
                weights["dst_grid_frac"].values if "dst_grid_frac" in weights else None)
        op = weights_matrix
        n_batch = int(np.prod(kept_shape)) if kept_shape else 1
        masked = bool(np.asarray(masked).any()) if np.ndim(masked) else bool(masked)

        src = source_data.data
        area_min, out_dtype = self.remap_area_min, self.out_dtype
        sb_in = isinstance(src, DeviceArray) and src.layout == "sb"
        if sb_in:
            n_h = len(source_data.dims) - len(kept_dims)
            if any(d not in horizontal_dims for d in source_data.dims[:n_h]):
                raise ValueError("a batch-fastest field (layout='sb') carries its horizontal dimensions first, "
                                 f"got dims {tuple(source_data.dims)}")
        elif self.keep_batch_fastest:
            raise ValueError("keep_batch_fastest needs a device-resident batch-fastest field "
                             "(source_data.data = DeviceArray(..., layout='sb'))")
        sb_out = sb_in and self.keep_batch_fastest
        out_shape = (tgt_shape + kept_shape) if sb_out else (kept_shape + tgt_shape)
        out_dims = (tgt_dims + kept_dims) if sb_out else (kept_dims + tgt_dims)

        def apply_rows(host):
            """(rows, S) host block -> (rows, D): chunks stream through the library's H2D / kernel /
            D2H pipeline."""
            host = np.asarray(host)
            if host.dtype not in (np.float32, np.float64):
                host = host.astype(np.float64)  # result_type(x, f64), regrid.py:550
            host = np.ascontiguousarray(host)
            if host.shape[1] != op.n_src:
                raise ValueError(f"source grid has {host.shape[1]} cells, weights expect {op.n_src}")
            return op.apply_host(host, masked=masked, remap_area_min=area_min, out_dtype=out_dtype)

        def compute():
            if sb_in:
                x = src.reshape(-1, n_batch)                  # (S, B): the batch values of a cell contiguous
                if x.shape[0] != op.n_src:
                    raise ValueError(f"source grid has {x.shape[0]} cells, weights expect {op.n_src}")
                y = op.apply(x, masked=masked, remap_area_min=area_min, out_dtype=out_dtype,
                             keep_batch_fastest=sb_out)
                return y.reshape(*out_shape)
            if isinstance(src, DeviceArray):
                x = src.reshape(n_batch, -1)
                if x.shape[1] != op.n_src:
                    raise ValueError(f"source grid has {x.shape[1]} cells, weights expect {op.n_src}")
                y = op.apply(x, masked=masked, remap_area_min=area_min, out_dtype=out_dtype)
                return y.reshape(*(kept_shape + tgt_shape))
            host = src.compute() if isinstance(src, LazyArray) else np.asarray(src)
            return apply_rows(host.reshape(n_batch, -1)).reshape(kept_shape + tgt_shape)

        if self.lazy and is_dask(src):
            # dask in, dask out: one task per block of the kept dimensions (regrid.py:538-541)
            out_data = map_batch_blocks(src, len(source_data.dims) - len(kept_dims), tgt_shape, apply_rows,
                                        dtype=out_dtype)
        elif self.lazy:
            out_data = LazyArray(out_shape, out_dtype, compute)
        else:
            out_data = compute()

        return self._finish(out_data, out_dims, source_data, kept_dims, weights,
                            tgt_shape, tgt_dims)

    # ------------------------------------------------------------------ apply (masked levels)
    def regrid3d(self, source_data, datagridtype):
        """regrid.py:339-427 as one grouped launch: per data level the nearest
        weights level (tolerance 1e-3) selects operator, mask and frac."""
        source_data = from_xarray(source_data)
        gridtype = self._get_gridtype(datagridtype)
        if gridtype is None:
            return DataArray(data=None)
        mask_dim = gridtype.mask_dim
        weights = gridtype.weights
        horizontal_dims = gridtype.horizontal_dims
        name = source_data.name or ''
        if "bnds" in name or "bounds" in name:
            return source_data
        if not any(x in source_data.dims for x in (horizontal_dims or [])):
            raise KeyError('Dimensions mismatch')

        wlev = np.asarray(weights.coords[mask_dim].values, dtype=np.float64)
        dlev = np.asarray(source_data.coords[mask_dim].values, dtype=np.float64)
        level_index = np.empty(dlev.size, dtype=np.int32)
        for idx, lev in enumerate(dlev):
            widx = int(np.argmin(np.abs(wlev - lev)))
            if not abs(wlev[widx] - lev) <= 1e-3:
                raise ValueError(f"{lev} not found in mask_dim {mask_dim}. "
                                 f"Available levels: {list(wlev)}")
            level_index[idx] = widx

        kept_dims = [d for d in source_data.dims if d not in horizontal_dims]
        p = kept_dims.index(mask_dim)
        outer_dims, inner_dims = kept_dims[:p], kept_dims[p + 1:]
        n_outer = int(np.prod([source_data.sizes[d] for d in outer_dims])) if outer_dims else 1
        n_inner = int(np.prod([source_data.sizes[d] for d in inner_dims])) if inner_dims else 1
        n_lev = source_data.sizes[mask_dim]
        rest_dims = outer_dims + inner_dims
        rest_shape = [source_data.sizes[d] for d in rest_dims]
        tgt_shape, tgt_dims = self._target_layout(weights)
        group = gridtype.group
        masked_levels = np.broadcast_to(np.asarray(gridtype.masked, dtype=bool),
                                        (len(group),)).astype(np.uint8)
        any_masked = bool(masked_levels.any())

        if self.transpose:
            out_dims = rest_dims + [mask_dim] + tgt_dims
            out_shape = rest_shape + [n_lev] + tgt_shape
        else:
            out_dims = [mask_dim] + rest_dims + tgt_dims
            out_shape = [n_lev] + rest_shape + tgt_shape

        src = source_data.data
        S, D = group.n_src, group.n_dst
        area_min, out_dtype, transpose = self.remap_area_min, self.out_dtype, self.transpose
        sb_in = isinstance(src, DeviceArray) and src.layout == "sb"
        if sb_in:
            # batch-fastest per level: (mask_dim, horizontal..., everything else...)
            n_h = len(source_data.dims) - len(kept_dims)
            if source_data.dims[0] != mask_dim or any(d not in horizontal_dims for d in source_data.dims[1:1 + n_h]):
                raise ValueError("a batch-fastest masked field (layout='sb') is laid out (mask_dim, horizontal "
                                 f"dims..., other dims...), got dims {tuple(source_data.dims)}")
        elif self.keep_batch_fastest:
            raise ValueError("keep_batch_fastest needs a device-resident batch-fastest field "
                             "(source_data.data = DeviceArray(..., layout='sb'))")
        sb_out = sb_in and self.keep_batch_fastest
        if sb_out:
            out_dims = [mask_dim] + tgt_dims + rest_dims
            out_shape = [n_lev] + tgt_shape + rest_shape

        def compute():
            if sb_in:
                x = src.reshape(n_lev, -1, n_outer * n_inner)
                if x.shape[1] != S:
                    raise ValueError(f"source grid has {x.shape[1]} cells, weights expect {S}")
                y = group.apply_sb(x, level_index, masked_levels, masked=any_masked, remap_area_min=area_min,
                                   transpose=transpose, out_dtype=out_dtype, keep_batch_fastest=sb_out)
                return y.reshape(*out_shape)
            if isinstance(src, DeviceArray):
                x = src.reshape(n_outer, n_lev, n_inner, -1)
                y = group.apply(x, level_index, masked_levels, masked=any_masked, remap_area_min=area_min,
                                transpose=transpose, out_dtype=out_dtype)
                return y.reshape(*out_shape)
            host = src.compute() if isinstance(src, LazyArray) else np.asarray(src)   # a dask field is computed here
            if host.dtype not in (np.float32, np.float64):
                host = host.astype(np.float64)
            host = np.ascontiguousarray(host).reshape(n_outer, n_lev, n_inner, -1)
            if host.shape[3] != S:
                raise ValueError(f"source grid has {host.shape[3]} cells, weights expect {S}")
            # host field: chunks of the outer axis stream through the group's pipeline
            out = group.apply_host(host, level_index, masked_levels, masked=any_masked, remap_area_min=area_min,
                                   transpose=transpose, out_dtype=out_dtype)
            return out.reshape(out_shape)

        out_data = LazyArray(out_shape, out_dtype, compute) if self.lazy else compute()

        kept_for_coords = kept_dims
        w2d = weights
        return self._finish(out_data, out_dims, source_data, kept_for_coords, w2d, tgt_shape, tgt_dims)


def regrid(source_data, target_grid=None, weights=None, transpose=True, cdo='cdo'):
    """One-shot helper (regrid.py:656-677)."""
    regridder = Regridder(source_data, target_grid=target_grid, weights=weights, cdo=cdo,
                          transpose=transpose)
    return regridder.regrid(source_data)
