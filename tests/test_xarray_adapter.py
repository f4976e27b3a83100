"""The xarray adapter of the facade (xrlite.is_xarray / from_xarray / to_xarray; reference
regrid.py:251-271) run against a stand-in `xarray` module -- the real package is absent from the
image, so these branches had never executed."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_xarray")


def run(code, timeout=300):
    env = dict(os.environ, PYTHONPATH=FAKE + os.pathsep + ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=timeout)


def test_roundtrip_of_dataarray_and_dataset():
    out = run(r'''
import numpy as np, xarray
from smmregrid_amd import xrlite
from smmregrid_amd.lazy import LazyArray
assert xrlite.HAVE_XARRAY
lat, lon = np.linspace(-80, 80, 5), np.arange(0, 360, 45.0)
xa = xarray.DataArray(np.arange(80.0).reshape(2, 5, 8), dims=("time", "lat", "lon"),
                      coords={"time": np.arange(2), "lat": (("lat",), lat, {"units": "degrees_north"}), "lon": lon},
                      attrs={"units": "K"}, name="tas")
assert xrlite.is_xarray(xa) and not xrlite.is_xarray(np.zeros(3))
lite = xrlite.from_xarray(xa)
assert isinstance(lite, xrlite.DataArray) and lite.dims == ("time", "lat", "lon") and lite.name == "tas"
assert lite.attrs == {"units": "K"} and lite.coords["lat"].attrs == {"units": "degrees_north"}
assert np.array_equal(lite.values, xa.values) and np.array_equal(lite.coords["lon"].values, lon)
back = xrlite.to_xarray(lite)
assert isinstance(back, xarray.DataArray) and back.dims == xa.dims and back.name == "tas" and back.attrs == xa.attrs
assert np.array_equal(back.values, xa.values) and np.array_equal(back.coords["lat"].values, lat)
ds = xarray.Dataset({"tas": xa, "pr": xarray.DataArray(np.ones((2, 5, 8)), dims=("time", "lat", "lon"), name="pr")},
                    attrs={"source": "stub"})
lds = xrlite.from_xarray(ds)
assert isinstance(lds, xrlite.Dataset) and list(lds.data_vars) == ["tas", "pr"] and lds.attrs == {"source": "stub"}
assert "lat" in lds.coords
bds = xrlite.to_xarray(lds)
assert isinstance(bds, xarray.Dataset) and list(bds.data_vars) == ["tas", "pr"] and bds.attrs == {"source": "stub"}
# a deferred result without dask is computed when it is handed to xarray
calls = []
lz = xrlite.DataArray(LazyArray((2, 3), np.float64, lambda: calls.append(1) or np.ones((2, 3))), dims=("a", "b"))
xz = xrlite.to_xarray(lz)
assert isinstance(xz, xarray.DataArray) and xz.shape == (2, 3) and np.array_equal(xz.values, np.ones((2, 3)))
assert xrlite.to_xarray(xrlite.DataArray(data=None)).data is None     # regrid2d's "no such grid" result
print("adapter-ok")
''')
    assert out.returncode == 0 and "adapter-ok" in out.stdout, out.stderr[-3000:]


@pytest.mark.gpu
def test_regrid_returns_the_type_it_was_given(hip):
    out = run(r'''
import numpy as np, xarray
from oracle import oracle
from smmregrid_amd import CdoGenerate, Regridder, gridgen
g = gridgen.parse_grid("r48x24")
rng = np.random.default_rng(3)
x = 280.0 + rng.standard_normal((3, 24, 48))
xa = xarray.DataArray(x, dims=("time", "lat", "lon"), coords={"time": np.arange(3), "lat": g.lat, "lon": g.lon},
                      attrs={"units": "K"}, name="tas")
w = CdoGenerate("r48x24", "r24x12").weights(method="con")
rg = Regridder(weights=w, device=0)
out = rg.regrid(xa)                                   # regrid.py:251-271: xarray in, xarray out
assert isinstance(out, xarray.DataArray) and out.dims == ("time", "lat", "lon") and out.shape == (3, 12, 24)
assert out.attrs == {"units": "K"} and out.name == "tas"
csr = oracle.coo_to_csr_c(1152, 288, w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values)
ref = oracle.apply_c(csr, x.reshape(3, -1), False, None, w["dst_grid_frac"].values, 0.5)
assert np.array_equal(out.values.reshape(3, -1), ref)
ods = rg.regrid(xarray.Dataset({"tas": xa}))
assert isinstance(ods, xarray.Dataset) and np.array_equal(ods["tas"].values, out.values)
try:
    rg.regrid(np.zeros((3, 24, 48)))
except TypeError as e:
    assert "not a Xarray object" in str(e)           # regrid.py:271
else:
    raise AssertionError("TypeError expected")
print("regrid-xarray-ok")
''')
    assert out.returncode == 0 and "regrid-xarray-ok" in out.stdout, out.stderr[-3000:]
