"""Data fixture from a file of the reference's own test suite (data only, no code):

    /opt/conda/bin/python3.9 tests/golden/make_ref_data_fixtures.py

tests/data/ua-ipsl.nc (used by the reference's tests/basic_test.py:95-102) -> first time step of
`ua` (19 pressure levels x 143 x 144, float32, missing values below the topography) with its
coordinates, as a compressed .npz.  Read with h5py because the file is HDF5-based NetCDF-4.
"""
import os

import h5py
import numpy as np

SRC = "/root/reference/tests/data/ua-ipsl.nc"
HERE = os.path.dirname(os.path.abspath(__file__))

with h5py.File(SRC, "r") as f:
    ua = f["ua"][0]
    fill = f["ua"].attrs["_FillValue"][0]
    ua = np.where(ua == fill, np.float32(np.nan), ua).astype(np.float32)     # what xarray's decoding yields
    np.savez_compressed(os.path.join(HERE, "ua_ipsl_t0.npz"), ua=ua,
                        lat=f["lat"][:].astype(np.float64), lon=f["lon"][:].astype(np.float64),
                        plev=f["plev"][:].astype(np.float64))
with h5py.File("/root/reference/tests/data/2t-era5.nc", "r") as f:
    # tests/data/2t-era5.nc (speed-evaluation.ipynb, identity tests): 12 x 73 x 144 float32 on a 2.5-degree
    # grid that includes both poles
    np.savez_compressed(os.path.join(HERE, "2t_era5.npz"), t2m=f["2t"][...].astype(np.float32),
                        lat=f["lat"][:].astype(np.float64), lon=f["lon"][:].astype(np.float64),
                        time=f["time"][:].astype(np.float64))
from scipy.io import netcdf_file   # tas-healpix2.nc is classic NetCDF-3
with netcdf_file("/root/reference/tests/data/tas-healpix2.nc", "r", mmap=False) as f:
    # tests/data/tas-healpix2.nc (identity2d_test.py:14-18): a HEALPix field (nside 32, nested) with explicit cell
    # coordinates in RADIANS; first two time steps + the coordinates as written
    units = f.variables["lat"].units
    np.savez_compressed(os.path.join(HERE, "tas_healpix2.npz"), tas=f.variables["tas"][:2].astype(np.float32),
                        lat=f.variables["lat"][:].astype(np.float64), lon=f.variables["lon"][:].astype(np.float64),
                        units=np.array(units.decode() if isinstance(units, bytes) else str(units)))
print("missing per level:", np.isnan(ua).reshape(19, -1).sum(axis=1))
with h5py.File("/root/reference/tests/data/temp3d-fesom.nc", "r") as f:
    # tests/data/temp3d-fesom.nc (identity3d_test.py:28-32): an unstructured ocean mesh, 3140 nodes x 47 levels, with
    # the cell polygons in lon_bnds / lat_bnds (16 vertices, short polygons padded with their last vertex); three of
    # the levels (the file holds 0, not a missing value, below the sea floor) + the mesh
    lev = [0, 23, 40]
    np.savez_compressed(os.path.join(HERE, "temp3d_fesom.npz"), temp=f["temp"][0][lev].astype(np.float32),
                        nz1=f["nz1"][:][lev].astype(np.float64),
                        lon=f["lon"][:].astype(np.float64), lat=f["lat"][:].astype(np.float64),
                        lon_bnds=f["lon_bnds"][...].astype(np.float32), lat_bnds=f["lat_bnds"][...].astype(np.float32))
# tests/data/lsm-ifs.grb (identity2d_test.py:22-27): the one GRIB file of the reference's tests, a data file, copied as it is
import shutil
shutil.copy("/root/reference/tests/data/lsm-ifs.grb", os.path.join(HERE, "grib", "lsm-ifs.grb"))
with h5py.File("/root/reference/tests/data/tas-ecearth.nc", "r") as f:
    # tests/data/tas-ecearth.nc (identity2d_test.py:30-53): EC-Earth near-surface temperature on the regular Gaussian
    # grid N128 (256 x 512) with lat_bnds / lon_bnds; two of the twelve months + the grid as the file holds it
    np.savez_compressed(os.path.join(HERE, "tas_ecearth.npz"), tas=f["tas"][:2].astype(np.float32),
                        lat=f["lat"][:].astype(np.float64), lon=f["lon"][:].astype(np.float64),
                        lat_bnds=f["lat_bnds"][...].astype(np.float64), lon_bnds=f["lon_bnds"][...].astype(np.float64))
