"""CPU tests of the host-side logic that needs no device: weight generation
(gridgen), labelled containers, dimension classification, weights IO."""
import os

import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import gridgen
from smmregrid_amd import io as smm_io
from smmregrid_amd.gridtype import GridType, tolist
from smmregrid_amd.xrlite import DataArray, Dataset


def test_cdo_grid_names():
    g = gridgen.parse_grid("r360x180")
    assert g.size == 64800 and g.lon[0] == 0.0 and g.lat[0] == -89.5 and g.lat[-1] == 89.5
    assert gridgen.parse_grid("r1440x721").size == 1038240
    hp = gridgen.parse_grid("hp32")
    assert hp.kind == "points" and hp.size == 12288          # basic_test.py:86 'hp32' -> 12288 cells
    with pytest.raises(ValueError):
        gridgen.parse_grid("gme30")                            # icosahedral: needs cdo


def test_healpix_centres_known_values():
    lon, lat = gridgen.healpix_centers(4, nested=True)
    np.testing.assert_allclose([lon[0], lat[0]], [45.0, 9.594068226860461], rtol=1e-12)
    lonr, latr = gridgen.healpix_centers(4, nested=False)
    np.testing.assert_allclose([lonr[0], latr[0]], [45.0, 78.28414760510762], rtol=1e-12)
    # nested and ring enumerate the same set of centres, equal-area pixels cover the sphere
    a = np.sort(np.round(lon * 1e6) * 1e3 + np.round(lat * 1e6) * 1e-3)
    b = np.sort(np.round(lonr * 1e6) * 1e3 + np.round(latr * 1e6) * 1e-3)
    np.testing.assert_allclose(a, b)
    assert abs(np.sin(np.radians(lat)).mean()) < 1e-12


@pytest.mark.parametrize("method,links", [("bil", 4), ("nn", 1)])
def test_point_weights_structure(method, links):
    w = gridgen.generate_weights("r72x36", "r24x12", method=method)
    D = w.sizes["dst_grid_size"]
    assert w.sizes["num_links"] == links * D
    src, dst, rm = w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values[:, 0]
    assert src.min() >= 1 and src.max() <= w.sizes["src_grid_size"] and dst.min() == 1 and dst.max() == D
    assert (np.diff(dst) >= 0).all()                          # sorted by destination like CDO
    np.testing.assert_allclose(np.bincount(dst - 1, weights=rm), 1.0, rtol=1e-13)
    assert (rm >= 0).all()
    assert w.attrs["source_grid"] == "lonlat" and w["dst_grid_dims"].values.tolist() == [24, 12]


def test_conservative_weights_with_mask_and_frac(rng):
    src = gridgen.parse_grid("r72x36")
    mask = (rng.random(src.size) > 0.3).astype(np.int32)
    w = gridgen.conservative_weights(src, "r24x12", src_mask=mask)
    src_a, dst_a, rm = w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values[:, 0]
    assert mask[src_a - 1].all()                              # no link from a masked source cell
    frac = w["dst_grid_frac"].values
    rows = np.bincount(dst_a - 1, weights=rm, minlength=frac.size)
    np.testing.assert_allclose(rows[frac > 0], 1.0, rtol=1e-12)   # fracarea normalisation
    assert (rows[frac == 0] == 0).all()
    assert 0.6 < frac.mean() < 0.8
    # destarea normalisation: rows sum to frac
    wd = gridgen.conservative_weights(src, "r24x12", src_mask=mask, norm="destarea")
    rows_d = np.bincount(wd["dst_address"].values - 1, weights=wd["remap_matrix"].values[:, 0],
                         minlength=frac.size)
    np.testing.assert_allclose(rows_d, frac, rtol=1e-12, atol=1e-15)
    # the cached-geometry path gives the same links
    lv = gridgen.ConservativeLevels(src, "r24x12").level(mask)
    assert np.array_equal(lv["src_address"].values, src_a)
    np.testing.assert_allclose(lv["remap_matrix"].values[:, 0], rm, rtol=1e-14)


def test_stacked_level_weights_layout(rng):
    # cdogenerate.py:310-343: zero-padded link arrays + link_length, per-level masks/frac
    src = gridgen.parse_grid("r48x24")
    masks = gridgen.synthetic_ocean_masks(48, 24, 3, top=0.8, bottom=0.3)
    assert (masks[1:] <= masks[:-1]).all()                    # deeper oceans nested in shallower ones
    w3 = gridgen.ConservativeLevels(src, "r12x6").stack(masks, [1.0, 10.0, 100.0])
    ll = w3["link_length"].values
    assert w3["src_address"].shape == (3, ll.max()) and (np.diff(ll) <= 0).all()
    for i in range(3):
        assert (w3["src_address"].values[i, ll[i]:] == 0).all()
        assert (w3["remap_matrix"].values[i, ll[i]:] == 0).all()
    assert w3["dst_grid_frac"].shape == (3, 72) and w3["src_grid_imask"].shape == (3, 48 * 24)
    assert list(w3.coords) == ["lev"] and w3["dst_grid_center_lat"].shape == (72,)


def test_gridtype_dimension_roles():
    gt = GridType(dims=("time", "lev", "lat", "lon"))
    assert gt.horizontal_dims == ["lat", "lon"] and gt.mask_dim == "lev" and gt.time_dims == ["time"]
    assert GridType(("time", "plev", "cell")).other_dims == ["plev"]
    assert GridType(("time", "plev", "cell"), extra_dims={"mask": ["plev"]}).mask_dim == "plev"
    with pytest.raises(ValueError):
        GridType(("lev", "depth", "lat"))                     # gridtype.py:155-157
    assert GridType(("lat", "lon")) == GridType(("time", "lon", "lat"))
    assert tolist(None) is None and tolist("a") == ["a"] and tolist(("a", "b")) == ["a", "b"]


def test_lite_containers(rng):
    x = rng.standard_normal((3, 4, 5))
    da = DataArray(x, dims=("time", "lat", "lon"), coords={"time": [0, 1, 2], "lat": np.arange(4.0)},
                   attrs={"units": "K"}, name="t")
    sub = da.isel(time=1)
    assert sub.dims == ("lat", "lon") and np.array_equal(sub.values, x[1]) and "time" in sub.coords
    sub2 = da.isel(time=[0, 2])
    assert sub2.shape == (2, 4, 5) and sub2.coords["time"].values.tolist() == [0, 2]
    ds = Dataset({"t": da, "flag": (("time",), np.arange(3))}, attrs={"title": "x"})
    assert ds.sizes["lon"] == 5 and "t" in ds and ds["flag"].dims == ("time",)
    out = ds.map(lambda v: v)
    assert list(out.data_vars) == ["t", "flag"] and out.attrs == {"title": "x"}
    assert list(ds.drop_vars(["flag"]).data_vars) == ["t"]


def test_weights_file_roundtrip_npz_and_netcdf3(tmp_path, rng):
    w = gridgen.conservative_weights("r36x18", "r12x6")
    path = os.path.join(tmp_path, "w.npz")
    smm_io.save_weights(w, path)
    r = smm_io.open_weights(path)
    assert r.attrs["source_grid"] == "lonlat"
    for k in ("src_address", "dst_address", "remap_matrix", "dst_grid_frac", "dst_grid_dims"):
        assert np.array_equal(r[k].values, w[k].values) and r[k].dims == w[k].dims
    # NetCDF-3 classic, the format `cdo -f nc gen*` writes (read via scipy, no netCDF4 needed)
    from scipy.io import netcdf_file
    nc_path = os.path.join(tmp_path, "w.nc")
    with netcdf_file(nc_path, "w") as nc:
        nc.source_grid = "lonlat"
        nc.dest_grid = "lonlat"
        for d, n in w.sizes.items():
            nc.createDimension(d, n)
        for k, v in w.data_vars.items():
            var = nc.createVariable(k, v.values.dtype.newbyteorder(">").char if v.values.dtype.kind == "f"
                                    else "i", v.dims)
            var[...] = v.values
    r2 = smm_io.open_weights(nc_path)
    assert r2.attrs["source_grid"] == "lonlat" and r2.sizes["num_links"] == w.sizes["num_links"]
    csr_a = oracle.coo_to_csr(648, 72, w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values)
    csr_b = oracle.coo_to_csr(648, 72, r2["src_address"].values, r2["dst_address"].values, r2["remap_matrix"].values)
    assert all(np.array_equal(a, b) for a, b in zip(csr_a, csr_b))


def test_gaussian_and_other_cdo_grid_names():
    g = gridgen.parse_grid("F128")                              # basic_test.py:85: F128 -> 256 * 512 cells
    assert g.size == 256 * 512 and g.cdo_type == "gaussian"
    assert g.lat_b[0] == -90.0 and g.lat_b[-1] == 90.0 and (np.diff(g.lat) > 0).all()
    np.testing.assert_allclose(g.lat[-1], 89.46282157, atol=1e-7)
    assert gridgen.parse_grid("n32").size == 128 * 64
    assert gridgen.parse_grid("hpz3").size == 768
    assert gridgen.parse_grid("global_2.5").size == 144 * 72
    w = gridgen.generate_weights("F32", "r24x12", method="con")
    rows = np.bincount(w["dst_address"].values - 1, weights=w["remap_matrix"].values[:, 0])
    np.testing.assert_allclose(rows, 1.0, rtol=1e-12)
    np.testing.assert_allclose(w["dst_grid_frac"].values, 1.0, rtol=1e-12)
    wb = gridgen.generate_weights("F32", "hp4", method="bil")
    assert wb.sizes["num_links"] == 4 * 192 and wb.attrs["source_grid"] == "gaussian"


@pytest.mark.parametrize("method", ["con", "bil", "nn"])
def test_north_to_south_latitudes_give_the_same_regrid(rng, method):
    """ERA5-style files store latitudes north-to-south: weights are computed south-to-north and
    renumbered to the file's cell order, so the regridded field does not depend on the direction."""
    lon, lat = np.arange(0, 360, 5.0), np.arange(87.5, -90, -5.0)
    x = rng.standard_normal((lat.size, lon.size))
    x[:5, :20] = np.nan
    down = gridgen.regular_grid_from_centers(lon, lat)
    up = gridgen.regular_grid_from_centers(lon, lat[::-1])
    assert down.lat_descending and not up.lat_descending

    def regrid(w, field2d):
        csr = oracle.coo_to_csr(w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values,
                                w["dst_address"].values, w["remap_matrix"].values)
        return oracle.apply(csr, field2d.reshape(1, -1), False, None, w["dst_grid_frac"].values, 0.5)

    wd = gridgen.generate_weights(down, "r24x12", method=method, src_mask=np.isfinite(x).ravel())
    wu = gridgen.generate_weights(up, "r24x12", method=method, src_mask=np.isfinite(x[::-1]).ravel())
    a, b = regrid(wd, x), regrid(wu, x[::-1])
    assert np.array_equal(np.isnan(a), np.isnan(b))
    np.testing.assert_allclose(a[~np.isnan(a)], b[~np.isnan(b)], rtol=1e-12, atol=1e-14)
    assert (np.diff(wd["dst_address"].values) >= 0).all()
    # a north-to-south target: rows of the result come out in the target's own order
    tgt = gridgen.regular_grid_from_centers(np.arange(0, 360, 15.0), np.arange(82.5, -90, -15.0))
    wt = gridgen.generate_weights("r72x36", tgt, method=method)
    wr = gridgen.generate_weights("r72x36", "r24x12", method=method)
    xx = rng.standard_normal((1, 72 * 36))
    np.testing.assert_allclose(regrid(wt, xx).reshape(12, 24)[::-1], regrid(wr, xx).reshape(12, 24), rtol=1e-12)
    np.testing.assert_allclose(np.degrees(wt["dst_grid_center_lat"].values[:24]), 82.5)


def test_pointwise_generators_honour_the_source_mask(rng):
    """bil / nn with a land mask: no link may touch a masked source cell (cdo genbil / gennn exclude
    them); bilinear weights are renormalised over the valid corners; cells without a valid corner
    get no link; nn picks the nearest unmasked cell; src_grid_imask is written."""
    from smmregrid_amd import gridgen
    src, dst = "r72x36", "r40x20"
    mask = (rng.random(72 * 36) > 0.35).astype(np.int32)
    mask.reshape(36, 72)[8:20, 10:30] = 0
    for method in ("bil", "nn"):
        w = gridgen.generate_weights(src, dst, method=method, src_mask=mask)
        sa, da, wt = w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values[:, 0]
        assert np.array_equal(w["src_grid_imask"].values, mask)
        assert (mask[sa - 1] == 1).all()
        rows = np.bincount(da - 1, weights=wt, minlength=800)
        has = np.bincount(da - 1, minlength=800) > 0
        np.testing.assert_allclose(rows[has], 1.0, rtol=1e-12)
        assert (wt >= 0).all()
        if method == "nn":
            assert has.all() and sa.size == 800
        else:
            assert not has.all()                      # interior of the masked block
            full = gridgen.generate_weights(src, dst, method="bil")
            # where all four corners are valid the masked and unmasked weights coincide
            f_sa = full["src_address"].values.reshape(800, 4)
            all_valid = (mask[f_sa - 1] == 1).all(axis=1)
            keep = np.isin(da - 1, np.flatnonzero(all_valid))
            assert np.array_equal(sa[keep], f_sa[all_valid].ravel())
            np.testing.assert_array_equal(wt[keep], full["remap_matrix"].values[:, 0].reshape(800, 4)[all_valid].ravel())
    # north-to-south source files: the mask is given in file order
    g = gridgen.parse_grid(src)
    import copy
    gd = copy.copy(g)
    gd.lat_descending = True
    wd = gridgen.generate_weights(gd, dst, method="bil", src_mask=mask)
    assert np.array_equal(wd["src_grid_imask"].values, mask)
    assert (mask[wd["src_address"].values - 1] == 1).all()


def test_netcdf3_writer_round_trips(tmp_path, rng):
    """io.write_netcdf3 (what CdoGenerate hands to the cdo binary, cdogenerate.py:82-87): weights and
    fields read back unchanged through io.open_weights, NaN kept as missing values."""
    from smmregrid_amd import DataArray, gridgen, io
    w = gridgen.generate_weights("r16x8", "r8x4", method="con")
    p = str(tmp_path / "w.nc")
    io.write_netcdf3(w, p)
    w2 = io.open_weights(p)
    assert w2.attrs["map_method"] == w.attrs["map_method"]
    for k, v in w.variables.items():
        assert tuple(w2.variables[k].dims) == tuple(v.dims), k
        assert np.array_equal(w2.variables[k].values, v.values), k
    g = gridgen.parse_grid("r12x6")
    x = rng.standard_normal((2, 6, 12))
    x[0, 2, 3:5] = np.nan
    fld = DataArray(x, dims=("time", "lat", "lon"), coords={"time": np.arange(2), "lat": g.lat, "lon": g.lon},
                    attrs={"units": "K"}, name="tos")
    p2 = str(tmp_path / "f.nc")
    io.write_netcdf3(fld, p2)
    f2 = io.open_dataset(p2)
    assert f2["tos"].dims == ("time", "lat", "lon") and f2["tos"].attrs["units"] == "K"
    assert np.array_equal(f2["tos"].values, x, equal_nan=True)
    assert np.array_equal(f2.coords["lat"].values, g.lat)
