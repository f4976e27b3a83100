"""CdoGenerate's `cdo` subprocess path (reference cdogenerate.py:234-303, :179-228) against a
stand-in binary: the command line, the environment, the read-back and the 3-D stacking layout."""
import json
import os

import numpy as np
import pytest

from smmregrid_amd import CdoGenerate, DataArray, gridgen
from tests.fake_cdo.make_fake_cdo import install

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def fake_cdo(tmp_path, monkeypatch):
    bindir = tmp_path / "bin"
    bindir.mkdir()
    install(str(bindir), ROOT)
    log = tmp_path / "cdo_calls.jsonl"
    monkeypatch.setenv("FAKE_CDO_LOG", str(log))
    monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ["PATH"])

    def calls():
        return [json.loads(line) for line in open(log)] if log.exists() else []
    return calls


def test_2d_command_line_environment_and_readback(fake_cdo, tmp_path):
    gen = CdoGenerate("r32x16", "r16x8", cdo_options=["-f", "nc"], cdo_extra=["-b", "F64"],
                      cdo_download_path=str(tmp_path))
    assert gen.have_cdo
    w = gen._cdo_weights("con", extrapolate=False, remap_norm="destarea", mask_dim=None, nproc=1)
    (call,) = fake_cdo()
    argv = call["argv"]
    # cdo [options] gen<method>,<target> [extra] <source> <weights file>   (cdogenerate.py:285-294)
    assert argv[:2] == ["-f", "nc"] and argv[2] == "gencon,r16x8" and argv[3:5] == ["-b", "F64"]
    assert argv[5] == "-const,1,r32x16"                       # CDO grid name as source (cdogenerate.py:90-92)
    assert argv[6].endswith(".nc") and len(argv) == 7
    assert call["REMAP_EXTRAPOLATE"] == "off" and call["CDO_REMAP_NORM"] == "destarea"   # :277-279
    assert call["CDO_DOWNLOAD_PATH"] == str(tmp_path)         # :63-64
    ref = gridgen.generate_weights("r32x16", "r16x8", method="con", norm="destarea")
    for k in ("src_address", "dst_address", "remap_matrix", "dst_grid_frac", "src_grid_imask", "dst_grid_dims"):
        assert np.array_equal(w[k].values, ref[k].values), k
    assert w.attrs["title"] == "fake cdo weights"             # it really is the file cdo wrote
    assert not os.path.exists(argv[6])                        # the temporary weights file is gone


def test_source_given_as_data_is_written_to_a_temporary_file(fake_cdo):
    g = gridgen.parse_grid("r24x12")
    x = np.ones((2, 12, 24))
    x[:, 3:6, 5:9] = np.nan
    field = DataArray(x, dims=("time", "lat", "lon"), coords={"time": np.arange(2), "lat": g.lat, "lon": g.lon},
                      name="tos")
    gen = CdoGenerate(field, "r12x6")
    w = gen._cdo_weights("con", True, "fracarea", None, 1)
    (call,) = fake_cdo()
    src_file = call["argv"][-2]
    assert src_file.endswith(".nc") and not os.path.exists(src_file)      # written, used, removed
    assert call["REMAP_EXTRAPOLATE"] == "on" and call["CDO_REMAP_NORM"] == "fracarea"
    mask = np.isfinite(x[0]).astype(np.int32).ravel()
    assert np.array_equal(w["src_grid_imask"].values, mask)               # the NaN mask reached cdo
    assert (mask[w["src_address"].values - 1] == 1).all()


def test_3d_one_run_per_level_and_stacked_layout(fake_cdo):
    g = gridgen.parse_grid("r24x12")
    L = 3
    x = np.ones((2, L, 12, 24))
    for lev in range(L):
        x[:, lev, 2:4 + 2 * lev, 3:8 + lev] = np.nan             # the mask grows with depth
    field = DataArray(x, dims=("time", "lev", "lat", "lon"),
                      coords={"time": np.arange(2), "lev": np.array([5.0, 50.0, 500.0]), "lat": g.lat, "lon": g.lon},
                      name="thetao")
    gen = CdoGenerate(field, "r12x6")
    w = gen._cdo_weights("con", True, "fracarea", "lev", nproc=2)
    calls = fake_cdo()
    assert len(calls) == L
    assert sorted(a for c in calls for a in c["argv"] if a.startswith("-sellevidx")) == \
        ["-sellevidx,1", "-sellevidx,2", "-sellevidx,3"]                   # cdogenerate.py:204
    # layout of cdogenerate.py:310-343: link arrays zero-padded to the longest level + link_length
    ll = w["link_length"].values
    assert ll.shape == (L,) and ll[0] > ll[1] > ll[2]
    assert w["src_address"].shape == (L, ll.max()) and w["remap_matrix"].shape == (L, ll.max(), 1)
    assert w["dst_grid_frac"].shape == (L, 72) and w["src_grid_imask"].shape == (L, 288)
    assert np.array_equal(w.coords["lev"].values, [5.0, 50.0, 500.0])
    for lev in range(L):
        assert (w["src_address"].values[lev, ll[lev]:] == 0).all()
        mask = np.isfinite(x[0, lev]).astype(np.int32).ravel()
        assert np.array_equal(w["src_grid_imask"].values[lev], mask)
    assert w["dst_grid_dims"].shape == (2,)                               # level-independent variables stay 2-D


def test_without_cdo_the_native_generator_says_so(monkeypatch, caplog):
    monkeypatch.setenv("PATH", "/nonexistent")
    gen = CdoGenerate("r32x16", "r16x8")
    assert not gen.have_cdo
    with pytest.raises(ValueError):
        gen.weights(method="bogus")
    with pytest.raises(ValueError):
        gen.weights(method="con", remap_norm="bogus")
    # every method the reference lists has a native form for lon/lat grids (round 4); a pair the native generator
    # cannot do still says so instead of guessing
    from smmregrid_amd.cdogenerate import NATIVE_METHODS
    assert set(NATIVE_METHODS) == {"bic", "bil", "con", "con2", "dis", "laf", "nn", "ycon"}     # cdogenerate.py:73
    for method in NATIVE_METHODS:
        w = gridgen.generate_weights("r32x16", "r16x8", method=method)          # (gen.weights adds the GPU mask pass)
        assert w.sizes["dst_grid_size"] == 16 * 8 and w.sizes["num_wgts"] == {"bic": 4, "con2": 3}.get(method, 1)
    with pytest.raises(ValueError, match="regular"):
        gridgen.generate_weights("r32x16", "hp4", method="con2")
    # REMAP_EXTRAPOLATE=off natively: target points outside the source grid get no link.  A regional source ...
    reg = gridgen.regular_grid_from_centers(np.arange(10.0, 60.0, 2.0), np.arange(-20.0, 31.0, 2.0))
    for method in ("bil", "bic", "nn", "dis"):
        on = gridgen.generate_weights(reg, "r36x18", method=method)
        off = gridgen.generate_weights(reg, "r36x18", method=method, extrapolate=False)
        linked_on, linked_off = np.unique(on["dst_address"].values), np.unique(off["dst_address"].values)
        assert linked_on.size == 36 * 18 and 0 < linked_off.size < 60
        dst = gridgen.parse_grid("r36x18")
        tl, tp = dst.centers()
        edge = 0.0 if method in ("bil", "bic") else 1.0            # hull of the centres / of the cells
        want = (tl >= 10.0 - edge) & (tl <= 58.0 + edge) & (tp >= -20.0 - edge) & (tp <= 30.0 + edge)
        assert np.array_equal(np.flatnonzero(want) + 1, linked_off)
        sel = np.isin(on["dst_address"].values, linked_off)        # inside, the weights are the same
        assert np.array_equal(on["remap_matrix"].values[sel], off["remap_matrix"].values)
        # the weights Dataset itself says which cells are covered -- a destination that lost its
        # links has fraction 0 and mask 0, not the extrapolating case's 1
        frac, imask = off["dst_grid_frac"].values, off["dst_grid_imask"].values
        assert np.array_equal(np.flatnonzero(frac > 0) + 1, linked_off) and np.array_equal(np.flatnonzero(imask) + 1, linked_off)
        assert (on["dst_grid_frac"].values > 0).all()
    # ... and a global one only loses the bilinear rows beyond its first / last row of centres
    off = gridgen.generate_weights("r32x16", "r16x30", method="bil", extrapolate=False)     # rows at +-87 > +-84.4
    assert np.unique(off["dst_address"].values).size == 16 * 28
    assert np.unique(gridgen.generate_weights("r32x16", "r16x30", method="nn", extrapolate=False)["dst_address"].values).size == 16 * 30


@pytest.mark.gpu
def test_regridder_uses_cdo_when_present(hip, fake_cdo, rng):
    """Regridder(source, target) on a box that has `cdo`: weights come from the subprocess, the mask
    pre-compute and the apply run on the GPU, results match the oracle on those weights."""
    from oracle import oracle
    from smmregrid_amd import Regridder
    from tests.helpers import assert_same
    g = gridgen.parse_grid("r48x24")
    x = 280.0 + 10.0 * rng.standard_normal((3, 24, 48))
    x[:, 5:12, 10:22] = np.nan
    field = DataArray(x, dims=("time", "lat", "lon"), coords={"time": np.arange(3), "lat": g.lat, "lon": g.lon},
                      name="tos")
    rg = Regridder(source_grid=field, target_grid="r24x12", method="con", device=0)
    assert len(fake_cdo()) == 1 and fake_cdo()[0]["argv"][0] == "gencon,r24x12"
    out = rg.regrid(field)
    w = rg.grids[0].weights
    assert w.attrs["title"] == "fake cdo weights" and "dst_grid_masked" in w
    csr = oracle.coo_to_csr_c(w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values,
                              w["dst_address"].values, w["remap_matrix"].values)
    imask = oracle.mask_apply_c(csr, w["src_grid_imask"].values)
    ref = oracle.apply_c(csr, x.reshape(3, -1), oracle.check_mask(imask), imask, w["dst_grid_frac"].values, 0.5)
    assert_same(out.values.reshape(3, -1), ref, exact=True)


def test_areas_through_cdo_gridarea(fake_cdo):
    """cdogenerate.py:345-400: source areas with cdo_extra / cdo_options, target areas without."""
    gen = CdoGenerate("r32x16", "r16x8", cdo_options=["-P", "2"], cdo_extra=["-b", "F64"])
    a = gen.areas()
    (call,) = fake_cdo()
    assert call["argv"][:4] == ["-P", "2", "-f", "nc4"] and call["argv"][4] == "gridarea"
    assert call["argv"][5:7] == ["-b", "F64"] and call["argv"][7] == "-const,1,r32x16"
    assert a["cell_area"].shape == (16, 32) and a["cell_area"].attrs["units"] == "m2"
    np.testing.assert_allclose(a["cell_area"].values.sum(), 4 * np.pi * 6371000.0 ** 2, rtol=1e-12)
    t = gen.areas(target=True)
    assert fake_cdo()[1]["argv"][:3] == ["-f", "nc4", "gridarea"] and t["cell_area"].shape == (8, 16)
    with pytest.raises(TypeError):
        CdoGenerate("r32x16").areas(target=True)


def test_checker_needs_the_cdo_binary(monkeypatch):
    """checker.py of the reference compares with CDO: without the binary there is nothing to compare with."""
    from smmregrid_amd.checker import check_cdo_regrid, find_var
    from smmregrid_amd import Dataset
    monkeypatch.setenv("PATH", "/nonexistent")
    with pytest.raises(FileNotFoundError, match="cdo"):
        check_cdo_regrid(os.path.join(ROOT, "tests", "golden", "refdata", "2t-era5.nc"), "r36x18")
    ds = Dataset({"tas": DataArray(np.zeros((2, 3, 4)), dims=("time", "lat", "lon")),
                  "time_bnds": DataArray(np.zeros((2, 2)), dims=("time", "bnds")),
                  "orog": DataArray(np.zeros((3, 4)), dims=("lat", "lon"))})
    assert find_var(ds) == ["tas"]                                    # checker.py:10-21
    assert find_var(Dataset({"orog": ds["orog"]})) == ["orog"]


@pytest.mark.gpu
@pytest.mark.parametrize("method,init_method", [("con", "grids"), ("bil", "weights"), ("nn", "grids")])
def test_check_cdo_regrid_on_a_reference_file(hip, fake_cdo, method, init_method):
    """identity2d_test.py:57-62 (`test_lonlat`: 2t-era5.nc -> r360x180.nc, init by weights) and the init-by-grids
    form: CDO's remap (the stand-in: per-step weights applied with scipy) against the GPU regrid, through the same
    function the reference's tests call."""
    from smmregrid_amd.checker import check_cdo_regrid
    golden = os.path.join(ROOT, "tests", "golden", "refdata")
    assert check_cdo_regrid(os.path.join(golden, "2t-era5.nc"), os.path.join(golden, "r360x180.nc"),
                            remap_method=method, init_method=init_method) is True
    calls = [c["argv"] for c in fake_cdo()]
    assert calls[0][0].startswith(f"remap{method},") and any(a[0].startswith(f"gen{method},") or
                                                             any(x.startswith(f"gen{method},") for x in a) for a in calls)
    with pytest.raises(KeyError):
        check_cdo_regrid(os.path.join(golden, "2t-era5.nc"), "r36x18", init_method="nothing")


@pytest.mark.gpu
def test_check_cdo_regrid_levels_and_masks(hip, fake_cdo, tmp_path, rng):
    """levels_test.py:10-27 / identity3d_test.py: a 3-D ocean-like file with a land mask growing with depth; all
    levels through check_cdo_regrid, levels [1, 3] and [2] through check_cdo_regrid_levels."""
    from smmregrid_amd import io
    from smmregrid_amd.checker import check_cdo_regrid, check_cdo_regrid_levels
    g = gridgen.parse_grid("r48x24")
    x = (10.0 + rng.standard_normal((2, 4, 24, 48))).astype(np.float32)
    for lev in range(4):
        x[:, lev, 4:8 + 3 * lev, 5:12 + 4 * lev] = np.nan
    field = DataArray(x, dims=("time", "lev", "lat", "lon"),
                      coords={"time": np.arange(2.0), "lev": np.array([5.0, 50.0, 500.0, 2000.0]), "lat": g.lat,
                              "lon": g.lon}, name="so")
    path = str(tmp_path / "so3d.nc")
    io.write_netcdf3(field, path)
    assert check_cdo_regrid(path, "r24x12", remap_method="con") is True
    assert check_cdo_regrid(path, "r24x12", remap_method="con", access="DataArray") is True
    assert check_cdo_regrid_levels(path, "r24x12", "lev", [1, 3], remap_method="con") is True
    assert check_cdo_regrid_levels(path, "r24x12", "lev", [2], remap_method="con") is True
    # the comparison notices a difference: a cut CDO did not apply
    assert check_cdo_regrid(path, "r24x12", remap_method="con", remap_area_min=0.9) is False


def test_vertical_dim_is_a_deprecated_alias_of_mask_dim(fake_cdo):
    """cdogenerate.py:156 / basic_test.py:104-110: `vertical_dim` still works and warns."""
    g = gridgen.parse_grid("r24x12")
    x = np.ones((1, 2, 12, 24))
    x[:, 1, 2:5, 3:9] = np.nan
    field = DataArray(x, dims=("time", "lev", "lat", "lon"),
                      coords={"time": np.arange(1), "lev": np.array([5.0, 50.0]), "lat": g.lat, "lon": g.lon}, name="so")
    gen = CdoGenerate(field, "r12x6")
    gen._with_masked_flag = lambda ds, mask_dim: ds            # the mask pass needs the GPU; not the subject here
    with pytest.warns(DeprecationWarning, match="vertical_dim"):
        w = gen.weights(method="nn", vertical_dim="lev")
    assert "lev" in w.sizes and w.sizes["lev"] == 2             # "Vertical coordinate 'lev' ... in weights dimensions"


def test_fields_on_curvilinear_and_unstructured_grids_reach_cdo_with_their_coordinates(fake_cdo, tmp_path):
    """cdogenerate.py:82-87: a source given as data is written to a temporary NetCDF file for cdo.  CDO finds the grid of
    a curvilinear / unstructured field through the CF `coordinates` attribute and the bounds the coordinates name: the
    file written here carries both (as xarray.to_netcdf's does), and reading it back gives the same grid."""
    from smmregrid_amd import Dataset, io
    from tests.test_gridgen_curvilinear import rotated_pole_grid
    lon, lat, clon, clat = rotated_pole_grid(nx=12, ny=6)
    nav_lon = DataArray(lon, dims=("y", "x"), attrs={"bounds": "bounds_nav_lon", "units": "degrees_east"})
    nav_lat = DataArray(lat, dims=("y", "x"), attrs={"bounds": "bounds_nav_lat", "units": "degrees_north"})
    ds = Dataset({"tos": DataArray(np.ones((2,) + lon.shape), dims=("time", "y", "x"),
                                   coords={"time": np.arange(2.0), "nav_lon": nav_lon, "nav_lat": nav_lat}, name="tos"),
                  "bounds_nav_lon": DataArray(clon, dims=("y", "x", "nvertex")),
                  "bounds_nav_lat": DataArray(clat, dims=("y", "x", "nvertex"))})
    path = str(tmp_path / "curv.nc")
    io.write_netcdf3(ds, path)
    back = io.open_dataset(path)
    assert back["tos"].attrs["coordinates"].split() == ["nav_lon", "nav_lat"] and "nav_lon" in back["tos"].coords
    g = CdoGenerate._grid_of(back)
    assert g.shape2d == (14, 6) and g.vertices[0].shape == (84, 4)
    np.testing.assert_allclose(g.vertices[1], clat.reshape(-1, 4))
    # the subprocess path: cdo (the stand-in reads the temporary file) sees the curvilinear cells and their corners
    gen = CdoGenerate(ds, "r24x12")
    w = gen._cdo_weights("con", True, "fracarea", None, 1)
    assert w["src_grid_dims"].values.tolist() == [14, 6] and w.sizes["dst_grid_size"] == 24 * 12
    ref = gridgen.generate_weights(CdoGenerate._grid_of(ds), "r24x12", method="con")
    assert np.array_equal(w["src_address"].values, ref["src_address"].values)
    np.testing.assert_allclose(w["remap_matrix"].values, ref["remap_matrix"].values, atol=1e-12)
