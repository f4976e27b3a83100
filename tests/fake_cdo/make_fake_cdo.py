"""Writes an executable stand-in for the `cdo` binary into a directory (tests only).

The stand-in logs its argv and the remap environment as one JSON line per call
($FAKE_CDO_LOG) and answers `gen<method>,<target> [-sellevidx,<k>] <source> <out>` with weights
from this package's own generator, written as NetCDF-3 -- enough to exercise the command line,
the environment and the read-back of CdoGenerate's subprocess path (cdogenerate.py:234-303 of
the reference) on a box without CDO."""
import os
import stat
import sys

SCRIPT = r'''#!{python}
import json, os, sys
sys.path.insert(0, {root!r})
import numpy as np
from smmregrid_amd import gridgen, io
from smmregrid_amd.cdogenerate import CdoGenerate
from smmregrid_amd.gridtype import GridType
from smmregrid_amd.xrlite import DataArray, Dataset

argv = sys.argv[1:]
with open(os.environ["FAKE_CDO_LOG"], "a") as f:
    f.write(json.dumps({{"argv": argv, "REMAP_EXTRAPOLATE": os.environ.get("REMAP_EXTRAPOLATE"),
                        "CDO_REMAP_NORM": os.environ.get("CDO_REMAP_NORM"),
                        "CDO_DOWNLOAD_PATH": os.environ.get("CDO_DOWNLOAD_PATH")}}) + "\n")
if "gridarea" in argv:                       # cdo [options] -f nc4 gridarea [extra] <grid> <out>
    out, source = argv[-1], argv[-2]
    g = gridgen.parse_grid(source.split(",", 2)[2]) if source.startswith("-const,1,") else \
        CdoGenerate._grid_of(next(v for v in io.open_dataset(source).data_vars.values()
                                  if GridType(v.dims).horizontal_dims))
    r = 6371000.0
    area = (np.diff(np.sin(np.radians(g.lat_b)))[:, None] * np.radians(np.diff(g.lon_b))[None, :]) * r * r
    io.write_netcdf3(Dataset({{"cell_area": (("lat", "lon"), area)}}, coords={{"lat": g.lat, "lon": g.lon}}), out)
    sys.exit(0)
gen = next(a for a in argv if a.startswith("gen"))
method, target = gen[3:].split(",", 1)
out, source = argv[-1], argv[-2]
lev = next((int(a.split(",")[1]) - 1 for a in argv if a.startswith("-sellevidx,")), None)
mask = None
if source.startswith("-const,1,"):
    src = gridgen.parse_grid(source.split(",", 2)[2])
else:
    ds = io.open_dataset(source)
    fld = next(v for v in ds.data_vars.values() if GridType(v.dims).horizontal_dims)
    src = CdoGenerate._grid_of(fld)
    gt = GridType(fld.dims)
    sel = {{d: 0 for d in fld.dims if d not in gt.horizontal_dims}}
    if lev is not None:
        levdim = [d for d in fld.dims if d not in gt.horizontal_dims and d not in (gt.time_dims or [])][0]
        sel[levdim] = lev
    v = fld.isel(**sel).values
    if not np.isfinite(v).all():
        mask = np.isfinite(v).astype(np.int32).ravel()
if os.path.isfile(target):
    tds = io.open_dataset(target)
    dst = CdoGenerate._grid_of(next(v for v in tds.data_vars.values() if GridType(v.dims).horizontal_dims))
else:
    dst = gridgen.parse_grid(target)
w = gridgen.generate_weights(src, dst, method=method, src_mask=mask, norm=os.environ.get("CDO_REMAP_NORM", "fracarea"))
w.attrs["title"] = "fake cdo weights"
io.write_netcdf3(w, out)
'''


def install(directory, root):
    path = os.path.join(directory, "cdo")
    with open(path, "w") as f:
        f.write(SCRIPT.format(python=sys.executable, root=root))
    os.chmod(path, os.stat(path).st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
    return path
