"""Deferred results (Regridder(lazy=True)) and the dask adapter (reference: lazy dask results with
the kept dimensions chunked, regrid.py:29-30, :538-541)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from smmregrid_amd.lazy import LazyArray

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lazy_array_defers_and_caches():
    calls = []

    def thunk():
        calls.append(1)
        return np.arange(24.0)

    a = LazyArray((2, 3, 4), np.float64, thunk)
    assert a.shape == (2, 3, 4) and a.ndim == 3 and a.size == 24 and a.dtype == np.float64
    assert not a.computed and not calls                    # nothing ran yet
    r = a.reshape(6, 4)
    assert r.shape == (6, 4) and not calls
    assert np.asarray(a).shape == (2, 3, 4) and len(calls) == 1
    assert a.values is a.compute() and len(calls) == 1      # cached
    assert np.array_equal(r.values, np.arange(24.0).reshape(6, 4)) and len(calls) == 1


def test_dask_adapter_maps_the_apply_over_batch_blocks():
    """`map_batch_blocks` under an interpreter that has dask, with the apply call injected (a scipy
    CSR product stands where the GPU apply goes): nothing runs before compute, the kept dimensions
    keep their chunking, the horizontal dimensions are rechunked to one block, values are right."""
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py) or subprocess.run([py, "-c", "import dask.array, scipy.sparse"],
                                                capture_output=True).returncode:
        pytest.skip("no interpreter with dask on this box")
    code = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, dask.array as da, scipy.sparse as sp
from smmregrid_amd.lazy import map_batch_blocks, is_dask
rng = np.random.default_rng(0)
nlat, nlon, D = 12, 20, 35
W = sp.random(D, nlat * nlon, density=0.05, random_state=1, format="csr")
calls = []
def apply_2d(block):                     # what Regridder hands in is op.apply_host
    assert block.ndim == 2 and block.shape[1] == nlat * nlon
    calls.append(block.shape[0])
    return (W @ block.T).T
x = rng.standard_normal((10, 3, nlat, nlon))
dx = da.from_array(x, chunks=(4, 2, 5, 7))          # horizontal dims chunked too: must be merged
assert is_dask(dx) and not is_dask(x)
y = map_batch_blocks(dx, 2, (5, 7), apply_2d)
assert is_dask(y) and y.shape == (10, 3, 5, 7) and y.dtype == np.float64
assert y.chunks == ((4, 4, 2), (2, 1), (5,), (7,)), y.chunks   # kept chunking preserved
assert calls == []                                   # lazy: nothing launched yet
got = y.compute(scheduler="threads")
ref = (W @ x.reshape(30, -1).T).T.reshape(10, 3, 5, 7)
assert np.allclose(got, ref, rtol=1e-13)
assert sorted(calls) == sorted([4 * 2, 4 * 1, 4 * 2, 4 * 1, 2 * 2, 2 * 1]), calls   # one call per kept block
# 1-D target (HEALPix / unstructured): new axis count differs from the dropped one
y1 = map_batch_blocks(dx, 2, (35,), apply_2d)
assert y1.shape == (10, 3, 35) and np.allclose(y1.compute(), ref.reshape(10, 3, 35))
print("dask-adapter-ok")
''' % ROOT
    out = subprocess.run([py, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "dask-adapter-ok" in out.stdout, out.stderr[-3000:]


@pytest.mark.gpu
def test_lazy_regridder_defers_the_launch(hip, rng):
    from oracle import oracle
    from smmregrid_amd import CdoGenerate, DataArray, Dataset, Regridder, gridgen
    from tests.helpers import assert_same
    g = gridgen.parse_grid("r96x48")
    x = (280.0 + 10.0 * rng.standard_normal((4, 48, 96))).astype(np.float32)
    field = DataArray(x, dims=("time", "lat", "lon"), coords={"time": np.arange(4), "lat": g.lat, "lon": g.lon},
                      name="tas")
    w = CdoGenerate("r96x48", "r36x18").weights(method="con")
    eager = Regridder(weights=w, device=0).regrid(field)
    lazy = Regridder(weights=w, device=0, lazy=True).regrid(field)
    assert isinstance(lazy.data, LazyArray) and not lazy.data.computed
    assert lazy.shape == eager.shape == (4, 18, 36) and lazy.dims == eager.dims
    assert list(lazy.coords) == list(eager.coords)          # metadata is available before the launch
    assert_same(lazy.values, eager.values, exact=True)
    assert lazy.data.computed
    ds = Regridder(weights=w, device=0, lazy=True).regrid(Dataset({"tas": field}))
    assert not ds["tas"].data.computed
    assert_same(ds["tas"].values, eager.values, exact=True)
    # masked levels
    L = 3
    x3 = 10.0 + rng.standard_normal((2, L, 48, 96))
    for lev in range(L):
        x3[:, lev, 10:20 + 4 * lev, 30:50] = np.nan
    f3 = DataArray(x3, dims=("time", "lev", "lat", "lon"),
                   coords={"time": np.arange(2), "lev": np.array([1.0, 2.0, 3.0]), "lat": g.lat, "lon": g.lon},
                   name="thetao")
    w3 = CdoGenerate(f3, "r36x18").weights(method="con", mask_dim="lev")
    e3 = Regridder(weights=w3, device=0).regrid(f3)
    l3 = Regridder(weights=w3, device=0, lazy=True).regrid(f3)
    assert isinstance(l3.data, LazyArray) and not l3.data.computed and l3.shape == e3.shape
    assert_same(l3.values, e3.values, exact=True)


@pytest.mark.gpu
def test_dask_field_through_the_regridder_on_the_gpu(hip):
    """Regridder(lazy=True) with a real dask array (the image's second interpreter has dask): the
    result is a dask array with the kept chunking, nothing is launched before compute, and the computed
    values are the eager ones (which the other tests tie to the oracle)."""
    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py) or subprocess.run([py, "-c", "import dask.array, scipy.sparse"],
                                                capture_output=True).returncode:
        pytest.skip("no interpreter with dask on this box")
    code = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, dask.array as da
from smmregrid_amd import CdoGenerate, DataArray, Regridder, gridgen
from smmregrid_amd.lazy import is_dask
g = gridgen.parse_grid("r96x48")
rng = np.random.default_rng(4)
x = (280.0 + 10.0 * rng.standard_normal((10, 48, 96))).astype(np.float32)
x[3, 10:20, 30:40] = np.nan
coords = {"time": np.arange(10), "lat": g.lat, "lon": g.lon}
w = CdoGenerate("r96x48", "r36x18").weights(method="con")
eager = Regridder(weights=w, device=0).regrid(DataArray(x, dims=("time", "lat", "lon"), coords=coords, name="tas"))
dx = da.from_array(x, chunks=(4, 24, 96))
lazy = Regridder(weights=w, device=0, lazy=True).regrid(DataArray(dx, dims=("time", "lat", "lon"), coords=coords,
                                                                  name="tas"))
assert is_dask(lazy.data) and lazy.shape == (10, 18, 36) and lazy.data.chunks == ((4, 4, 2), (18,), (36,))
got = lazy.data.compute(scheduler="threads")          # blocks call smm_apply_host concurrently: they take turns
ev = eager.values
assert got.dtype == np.float64 and np.array_equal(np.isnan(got), np.isnan(ev))
assert np.array_equal(got[~np.isnan(got)], ev[~np.isnan(ev)])
assert np.array_equal(np.asarray(lazy.values), got) or np.array_equal(np.isnan(lazy.values), np.isnan(got))
print("dask-gpu-ok")
''' % ROOT
    # that interpreter ships an older libstdc++ than the one the HIP library was linked against
    env = dict(os.environ)
    std = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"
    if os.path.exists(std):
        env["LD_PRELOAD"] = std + (":" + env["LD_PRELOAD"] if env.get("LD_PRELOAD") else "")
    out = subprocess.run([py, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    if "GLIBCXX" in out.stderr:
        pytest.skip("the dask interpreter cannot load the HIP library (libstdc++ too old)")
    assert out.returncode == 0 and "dask-gpu-ok" in out.stdout, (out.stdout + out.stderr)[-3000:]
