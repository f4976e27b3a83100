"""Comparison against CDO itself (mirror of the reference's smmregrid/checker.py, the function its whole test suite
is built on: tests/identity2d_test.py, identity3d_test.py, levels_test.py call nothing else).

`check_cdo_regrid(finput, ftarget, ...)` regrids `finput` onto the grid of `ftarget` twice -- with the `cdo` binary
(`cdo remap<method>,<ftarget> <finput> <out>`; the reference goes through the python-cdo wrapper, which runs the same
command line) and with `Regridder` on the GPU -- and answers whether the fields agree (`numpy.allclose`, NaN == NaN),
as checker.py:24-70 does.  `check_cdo_regrid_levels` is the variant for level sub-selection (checker.py:72-124).

It needs a `cdo` binary: on a box without one it raises FileNotFoundError (nothing here stands in for CDO)."""
import os
import shutil
import subprocess
import tempfile

import numpy as np

from .cdogenerate import CdoGenerate
from .regrid import Regridder
from .xrlite import Dataset, from_xarray


def find_var(xfield):
    """checker.py:10-21: the variables that most likely want regridding -- those with a time dimension and no
    bounds dimension, else all of them."""
    myvar = [name for name, v in xfield.data_vars.items() if 'time' in v.dims and 'bnds' not in v.dims]
    return myvar or list(xfield.data_vars)


def _cdo_remap(cdo, remap_method, ftarget, finput, env):
    """`cdo remap<method>,<ftarget> <finput> <tmp>` -> Dataset.  `finput` may be a Dataset (written to a temporary
    NetCDF file first, as the python-cdo wrapper does for xarray input)."""
    from .io import open_dataset, write_netcdf3
    if shutil.which(cdo) is None:
        raise FileNotFoundError(f"check_cdo_regrid compares with the '{cdo}' binary, which is not on this box")
    tmp_in = None
    if not isinstance(finput, str):
        fd, tmp_in = tempfile.mkstemp(suffix=".nc")
        os.close(fd)
        write_netcdf3(from_xarray(finput), tmp_in)
    fd, tmp_out = tempfile.mkstemp(suffix=".nc")
    os.close(fd)
    try:
        run_env = dict(os.environ)
        run_env.update(env)
        proc = subprocess.run([cdo, f"remap{remap_method},{ftarget}", tmp_in or finput, tmp_out], env=run_env,
                              capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"cdo remap{remap_method} failed: {proc.stderr.strip()}")
        return open_dataset(tmp_out)
    finally:
        for path in (tmp_in, tmp_out):
            if path and os.path.exists(path):
                os.remove(path)


def _stack(ds, names):
    """`ds[names].to_array()`: the chosen variables stacked along a new leading axis."""
    return np.stack([np.asarray(ds[name].values, dtype=np.float64) for name in names])


def _open(finput):
    from .io import open_dataset
    return open_dataset(finput) if isinstance(finput, str) else from_xarray(finput)


def check_cdo_regrid(finput, ftarget, remap_method='con', access='Dataset',
                     init_method='grids', mask_dim=None, vertical_dim=None, extrapolate=True,
                     remap_area_min=0.0, loglevel='INFO', cdo='cdo'):
    """checker.py:24-70 (same keywords, plus the name of the binary).  True when CDO's remap and the GPU regrid of
    the variables `find_var` picks agree."""
    xfield = _open(finput)
    cdofield = _cdo_remap(cdo, remap_method, ftarget, finput,
                          {'REMAP_EXTRAPOLATE': 'on' if extrapolate else 'off'})
    smmvar, cdovar = find_var(xfield), find_var(cdofield)
    if init_method == 'grids':
        interpolator = Regridder(source_grid=finput, target_grid=ftarget, remap_area_min=remap_area_min,
                                 method=remap_method, mask_dim=mask_dim, vertical_dim=vertical_dim,
                                 loglevel=loglevel, cdo=cdo)
    elif init_method == 'weights':
        wfield = CdoGenerate(finput, ftarget, loglevel=loglevel, cdo=cdo).weights(
            method=remap_method, mask_dim=mask_dim, vertical_dim=vertical_dim)
        interpolator = Regridder(weights=wfield, loglevel=loglevel, remap_area_min=remap_area_min)
    else:
        raise KeyError('Unsupported init method')
    rfield = interpolator.regrid(xfield)
    if access == 'Dataset':
        got, want = _stack(rfield, smmvar), _stack(cdofield, cdovar)
    else:
        got = np.asarray((rfield[smmvar[-1]] if isinstance(rfield, Dataset) else rfield).values, dtype=np.float64)
        want = np.asarray(cdofield[cdovar[-1]].values, dtype=np.float64)
    return bool(got.shape == want.shape and np.allclose(want, got, equal_nan=True))


def check_cdo_regrid_levels(finput, ftarget, mask_dim, levels, remap_method='con',
                            remap_area_min=0.5, access='Dataset',
                            extrapolate=True, loglevel='INFO', cdo='cdo'):
    """checker.py:72-124: CDO regrids every level, the GPU path the levels `levels` of `mask_dim` only, with the full
    3-D weights -- the level matching of regrid3d (regrid.py:386-395) has to pick the right operators."""
    xfield = _open(finput)
    cdofield = _cdo_remap(cdo, remap_method, ftarget, finput,
                          {'REMAP_EXTRAPOLATE': 'on' if extrapolate else 'off',
                           'REMAP_AREA_MIN': str(remap_area_min)})
    smmvar, cdovar = find_var(xfield), find_var(cdofield)
    wfield = CdoGenerate(finput, ftarget, loglevel=loglevel, cdo=cdo).weights(method=remap_method, mask_dim=mask_dim)
    interpolator = Regridder(weights=wfield, loglevel=loglevel, remap_area_min=remap_area_min)
    sub = Dataset({k: (v.isel(**{mask_dim: list(levels)}) if mask_dim in v.dims else v)
                   for k, v in xfield.data_vars.items()}, attrs=xfield.attrs)
    rfield = interpolator.regrid(sub)
    want_ds = {k: (v.isel(**{mask_dim: list(levels)}) if mask_dim in v.dims else v) for k, v in cdofield.data_vars.items()}
    if access == 'Dataset':
        got = _stack(rfield, smmvar)
        want = np.stack([np.asarray(want_ds[name].values, dtype=np.float64) for name in cdovar])
    else:
        got = np.asarray(rfield[smmvar[-1]].values, dtype=np.float64)
        want = np.asarray(want_ds[cdovar[-1]].values, dtype=np.float64)
    return bool(got.shape == want.shape and np.allclose(want, got, equal_nan=True))
