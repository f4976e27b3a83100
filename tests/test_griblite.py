"""GRIB edition-1 reader (smmregrid_amd/griblite.py).  The reference's tests read tests/data/lsm-ifs.grb through
cfgrib (identity2d_test.py:22-27, `nn`; util_test.py:57 expects the grid to be recognised as reduced Gaussian); there is no GRIB library on this image, so the decoder is pinned by what is KNOWN about that
file (an IFS land-sea mask on the octahedral reduced Gaussian grid O96) and by messages encoded here, in the test, from
the published section layout."""
import os

import numpy as np
import pytest

from smmregrid_amd import griblite
from smmregrid_amd.io import open_dataset

REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grib", "lsm-ifs.grb")


def test_the_reference_land_sea_mask_on_o96():
    ds = open_dataset(REF)
    lsm = ds["lsm"]
    assert lsm.dims == ("values",) and lsm.shape == (40320,) and lsm.values.dtype == np.float32
    assert lsm.attrs["GRIB_gridType"] == "reduced_gg" and lsm.attrs["units"] == "(0 - 1)"
    v = lsm.values
    assert v.min() == 0.0 and v.max() == 1.0 and np.isin(v, [0.0, 1.0]).mean() > 0.7     # a mask with fractional coasts
    m = griblite.read_messages(REF)[0]
    # the octahedral reduced Gaussian grid O96: 192 rows, 4 i + 16 points in row i from the pole, 400 at the equator
    assert m.pl.size == 192 and np.array_equal(m.pl[:96], 4 * np.arange(1, 97) + 16) and m.pl.max() == 400
    assert np.array_equal(m.pl, m.pl[::-1]) and m.pl.sum() == 40320
    lat, lon = ds.coords["latitude"].values, ds.coords["longitude"].values
    assert lat.shape == lon.shape == (40320,)
    assert lat[0] == pytest.approx(89.28423, abs=1e-4) and lat[-1] == pytest.approx(-89.28423, abs=1e-4)
    assert (np.diff(lat) <= 0).all() and lon[:20].tolist() == pytest.approx((np.arange(20) * 18.0).tolist())
    # area-weighted land share of the globe: 29 %
    _, wg = np.polynomial.legendre.leggauss(192)
    starts = np.cumsum(np.r_[0, m.pl[:-1]])
    land = sum(wg[j] / 2 * v[s:s + p].mean() for j, (s, p) in enumerate(zip(starts, m.pl)))
    assert land == pytest.approx(0.29, abs=0.01)
    # geography: the South Pole row is ice, the central Pacific is sea, central Asia is land
    def at(la, lo):
        return v[np.argmin((lat - la) ** 2 + (((lon - lo + 180) % 360) - 180) ** 2)]
    assert at(-89.0, 0.0) == 1.0 and at(0.0, 200.0) == 0.0 and at(45.0, 90.0) == 1.0 and at(-25.0, 135.0) == 1.0


def ibm32(x):
    """Encode a float as IBM hexadecimal single precision (exactly representable inputs only)."""
    if x == 0:
        return bytes(4)
    sign = 0x80 if x < 0 else 0
    x, e = abs(x), 64
    while x >= 1.0:
        x, e = x / 16.0, e + 1
    while x < 1.0 / 16.0:
        x, e = x * 16.0, e - 1
    return bytes([sign | e]) + int(round(x * (1 << 24))).to_bytes(3, "big")


def sm(value, nbytes):
    """Sign-and-magnitude integer."""
    return ((1 << (8 * nbytes - 1)) | -value if value < 0 else value).to_bytes(nbytes, "big")


def encode(values, rep, ni, nj, la1, lo1, la2, lo2, n_or_dj, param=167, level_type=1, level=0, date=(2020, 1, 1, 0),
           step=0, nbits=16, decimal=0, bitmap=None, pl=None, scan=0):
    """A GRIB-1 message in simple packing, written from the section layout (WMO FM 92, edition 1)."""
    vals = np.asarray(values, dtype=np.float64).ravel()
    present = vals if bitmap is None else vals[np.asarray(bitmap, bool).ravel()]
    scaled = present * 10.0 ** decimal
    ref = float(np.floor(scaled.min() * 16.0) / 16.0)
    span = scaled.max() - ref
    e = 0 if span == 0 else int(np.ceil(np.log2(span / ((1 << nbits) - 1))))
    x = np.round((scaled - ref) / 2.0 ** e).astype(np.uint64)
    bits = ((x[:, None] >> np.arange(nbits - 1, -1, -1, dtype=np.uint64)) & np.uint64(1)).astype(np.uint8).ravel()
    pad = (-bits.size) % 16
    data = np.packbits(np.r_[bits, np.zeros(pad, np.uint8)]).tobytes()
    y, mo, d, h = date
    pds = bytearray(28)
    pds[0:3] = (28).to_bytes(3, "big")
    pds[3], pds[4], pds[5], pds[6] = 128, 98, 1, 255
    pds[7] = 0x80 | (0x40 if bitmap is not None else 0)
    pds[8], pds[9] = param, level_type
    pds[10:12] = level.to_bytes(2, "big")
    pds[12], pds[13], pds[14], pds[15], pds[16] = y % 100 or 100, mo, d, h, 0
    pds[17], pds[18], pds[19], pds[20] = 1, step, 0, 0
    pds[24] = (y - 1) // 100 + 1
    pds[26:28] = sm(decimal, 2)
    gds = bytearray(32 + (2 * nj if pl is not None else 0))
    gds[0:3] = len(gds).to_bytes(3, "big")
    gds[3], gds[4], gds[5] = 0, (33 if pl is not None else 255), rep
    gds[6:8] = (0xFFFF if pl is not None else ni).to_bytes(2, "big")
    gds[8:10] = nj.to_bytes(2, "big")
    gds[10:13], gds[13:16] = sm(int(round(la1 * 1000)), 3), sm(int(round(lo1 * 1000)), 3)
    gds[16] = 0x80
    gds[17:20], gds[20:23] = sm(int(round(la2 * 1000)), 3), sm(int(round(lo2 * 1000)), 3)
    gds[23:25] = (0xFFFF).to_bytes(2, "big")
    gds[25:27] = int(n_or_dj).to_bytes(2, "big")
    gds[27] = scan
    if pl is not None:
        gds[32:] = np.asarray(pl, dtype=">u2").tobytes()
    bms = b""
    if bitmap is not None:
        bm = np.packbits(np.asarray(bitmap, np.uint8).ravel())
        body = bm.tobytes() + (b"\x00" if (6 + bm.size) % 2 else b"")
        bms = (6 + len(body)).to_bytes(3, "big") + bytes([0]) + (0).to_bytes(2, "big") + body
    bds = (11 + len(data)).to_bytes(3, "big") + bytes([pad & 0x0F]) + sm(e, 2) + ibm32(ref) + bytes([nbits]) + data
    body = bytes(pds) + bytes(gds) + bms + bds + b"7777"
    return b"GRIB" + (8 + len(body)).to_bytes(3, "big") + b"\x01" + body


def test_round_trip_lonlat_grid_levels_times_and_bitmap(tmp_path, rng):
    """A lon/lat file with two times x three pressure levels of temperature and a surface field with missing points."""
    ni, nj = 36, 19
    lat, lon = np.linspace(90, -90, nj), np.arange(ni) * 10.0
    msgs, want = [], np.empty((2, 3, nj, ni))
    for t, day in enumerate((1, 2)):
        for k, lev in enumerate((850, 500, 250)):
            f = 250.0 + 30.0 * np.cos(np.radians(lat))[:, None] + rng.standard_normal((nj, ni)) - 10.0 * k + t
            want[t, k] = f
            msgs.append(encode(f, 0, ni, nj, 90, 0, -90, 350, 10000, param=130, level_type=100, level=lev,
                               date=(2021, 3, day, 12), nbits=24))
    sst = 290.0 + rng.standard_normal((nj, ni))
    sea = rng.random((nj, ni)) > 0.3
    msgs.append(encode(sst, 0, ni, nj, 90, 0, -90, 350, 10000, param=34, bitmap=sea, date=(2021, 3, 1, 12), nbits=12))
    path = tmp_path / "synthetic.grib"
    path.write_bytes(b"".join(msgs))
    ds = open_dataset(str(path))
    t = ds["t"]
    assert t.dims == ("time", "isobaricInhPa", "latitude", "longitude") and t.shape == (2, 3, nj, ni)
    assert t.coords["isobaricInhPa"].values.tolist() == [250.0, 500.0, 850.0]              # sorted, as cfgrib sorts
    np.testing.assert_allclose(t.values[:, ::-1], want, atol=60.0 / (1 << 24) * 2 + 4e-5)  # 24-bit packing, f32 result
    np.testing.assert_allclose(ds.coords["latitude"].values, lat)
    np.testing.assert_allclose(ds.coords["longitude"].values, lon)
    assert np.diff(t.coords["time"].values).tolist() == [86400.0]
    s = ds["sst"]
    assert s.dims == ("latitude", "longitude") and np.array_equal(np.isnan(s.values), ~sea)
    np.testing.assert_allclose(s.values[sea], sst[sea], atol=8.0 / (1 << 12) + 4e-5)


def test_round_trip_gaussian_grids_regular_and_reduced(tmp_path, rng):
    n = 8
    lat = griblite.gaussian_latitudes(n)
    assert lat.size == 16 and np.allclose(lat, -lat[::-1])
    # first rows of the T42 (N32) and T106 (N80) grids, as every model description lists them
    assert griblite.gaussian_latitudes(32)[0] == pytest.approx(87.8638, abs=1e-4)
    assert griblite.gaussian_latitudes(80)[0] == pytest.approx(89.1416, abs=1e-4)
    reg = rng.random((16, 32))
    p1 = tmp_path / "regular_gg.grb"
    p1.write_bytes(encode(reg, 4, 32, 16, lat[0], 0, lat[-1], 348.75, n, param=172, decimal=0))
    ds = open_dataset(str(p1))
    assert ds["lsm"].dims == ("latitude", "longitude") and ds["lsm"].attrs["GRIB_gridType"] == "regular_gg"
    np.testing.assert_allclose(ds["lsm"].values, reg, atol=2.0 / (1 << 16))
    np.testing.assert_allclose(ds.coords["latitude"].values, lat)
    pl = np.array([8, 12, 16, 20, 24, 28, 32, 32, 32, 32, 28, 24, 20, 16, 12, 8])
    red = rng.random(pl.sum()) * 40.0 - 20.0
    p2 = tmp_path / "reduced_gg.grb"
    p2.write_bytes(encode(red, 4, 0, 16, lat[0], 0, lat[-1], 348.75, n, param=167, pl=pl, decimal=2))
    ds = open_dataset(str(p2))
    assert ds["t2m"].dims == ("values",) and ds["t2m"].shape == (pl.sum(),)
    np.testing.assert_allclose(ds["t2m"].values, red, atol=0.006)                          # decimal scale 2
    assert np.array_equal(ds.coords["latitude"].values, np.repeat(lat, pl))
    assert ds.coords["longitude"].values[8:20].tolist() == pytest.approx((np.arange(12) * 30.0).tolist())
    # south-to-north scanning
    p3 = tmp_path / "s2n.grb"
    p3.write_bytes(encode(reg[::-1], 4, 32, 16, lat[-1], 0, lat[0], 348.75, n, param=172, scan=0x40))
    ds3 = open_dataset(str(p3))
    np.testing.assert_allclose(ds3.coords["latitude"].values, lat[::-1])
    np.testing.assert_allclose(ds3["lsm"].values, reg[::-1], atol=2.0 / (1 << 16))


def test_what_is_not_decoded_says_so(tmp_path):
    good = encode(np.arange(12.0).reshape(3, 4), 0, 4, 3, 10, 0, -10, 30, 10000)
    bad = bytearray(good)
    bad[7] = 3
    p = tmp_path / "ed3.grb"
    p.write_bytes(bytes(bad))
    with pytest.raises(griblite.GribUnsupported, match="edition 3"):
        open_dataset(str(p))
    rot = bytearray(good)
    rot[8 + 28 + 5] = 10                                # rotated lon/lat
    p.write_bytes(bytes(rot))
    with pytest.raises(griblite.GribUnsupported, match="representation type 10"):
        open_dataset(str(p))
    cplx = bytearray(good)
    cplx[8 + 28 + 32 + 3] |= 0x40                       # second-order packing flag
    p.write_bytes(bytes(cplx))
    with pytest.raises(griblite.GribUnsupported, match="second-order"):
        open_dataset(str(p))
    p.write_bytes(good[:-10])
    with pytest.raises(ValueError):
        open_dataset(str(p))


def test_the_grid_of_the_grib_file_is_recognised_and_feeds_the_generator():
    """util_test.py:57: `lsm-ifs.grb` is a GaussianReduced grid; identity2d_test.py:22-27 regrids it with `nn`."""
    from smmregrid_amd import CdoGenerate, gridgen
    from smmregrid_amd.gridmeta import GridInspector
    ds = open_dataset(REF)
    (gt,) = GridInspector(ds).get_gridtype()
    assert gt.kind == "GaussianReduced" and gt.horizontal_dims == ["values"] and list(gt.variables) == ["lsm"]
    g = CdoGenerate._grid_of(ds)
    assert g.kind == "points" and g.size == 40320
    w = gridgen.generate_weights(g, "r72x36", method="nn")
    assert w.sizes["num_links"] == 72 * 36 and w["src_grid_dims"].values.tolist() == [40320]
    y = ds["lsm"].values[w["src_address"].values - 1].reshape(36, 72)
    assert y[0].min() == 1.0 and y[:, 40].mean() < 0.2 and 0.2 < y.mean() < 0.45     # Antarctica, the date line, the globe


def test_reduced_grid_cells_for_conservative_weights_and_areas():
    """A reduced Gaussian grid implies its cells (latitude bands of the Gaussian weights x equal arcs of longitude): the
    native generator builds them from the list of centres, so `con` and `areas()` work on the GRIB file as well
    (CDO does the same for gridtype gaussian_reduced)."""
    from smmregrid_amd import CdoGenerate, gridgen
    ds = open_dataset(REF)
    g = CdoGenerate._grid_of(ds)
    assert g.cdo_type == "gaussian_reduced" and g.vertices[0].shape[0] == 40320
    m = griblite.read_messages(REF)[0]
    _, wg = np.polynomial.legendre.leggauss(192)
    exact = np.repeat(2 * np.pi * wg[::-1] / m.pl, m.pl)                     # band area / points of the row
    got = gridgen.polygon_areas(*g.vertices)
    np.testing.assert_allclose(got, exact, rtol=5e-4)                         # parallels followed by <= 2-degree chords
    area = CdoGenerate(ds, "r360x180", cdo="no-such-cdo-binary").areas()["cell_area"].values
    assert area.shape == (40320,) and area.sum() / 1e6 == pytest.approx(5.101e8, rel=1e-3)      # areas_test.py:12
    w = gridgen.generate_weights(g, "r180x90", method="con")
    np.testing.assert_allclose(w["dst_grid_frac"].values, 1.0, atol=1e-12)   # no target sample falls between cells
    d, s = w["dst_address"].values - 1, w["src_address"].values - 1
    y = np.bincount(d, weights=w["remap_matrix"].values[:, 0] * ds["lsm"].values[s], minlength=180 * 90)
    land_exact = (np.repeat(wg[::-1] / 2 / m.pl, m.pl) * ds["lsm"].values).sum()
    assert (y * w["dst_grid_area"].values).sum() / (4 * np.pi) == pytest.approx(land_exact, abs=3e-4)
    # lists that are no reduced grid: scattered points, rows with uneven spacing
    rng = np.random.default_rng(1)
    assert gridgen.reduced_grid_vertices(rng.uniform(0, 360, 50), rng.uniform(-90, 90, 50)) is None
    lon = np.r_[0.0, 100.0, 200.0, 0.0, 90.0, 180.0, 270.0, 0.0, 120.0, 240.0]
    lat = np.r_[[60.0] * 3, [0.0] * 4, [-60.0] * 3]
    assert gridgen.reduced_grid_vertices(lon, lat) is None                    # first row is not evenly spaced
    lon[1:3] = [120.0, 240.0]
    lon_v, lat_v = gridgen.reduced_grid_vertices(lon, lat)                    # non-Gaussian rows: mid-point bands
    assert lat_v[0].min() == 30.0 and lat_v[0].max() == 90.0 and lat_v[4].min() == -30.0 and lat_v[-1].min() == -90.0
    np.testing.assert_allclose(gridgen.polygon_areas(lon_v, lat_v).sum(), 4 * np.pi, rtol=2e-2)


# ---------------------------------------------------------------------------------------------- edition 2
def section(number, body):
    return (5 + len(body)).to_bytes(4, "big") + bytes([number]) + bytes(body)


def encode2(fields, template=0, ni=0, nj=0, la1=0.0, lo1=0.0, la2=0.0, lo2=0.0, n_or_dj=0, pl=None, scan=0,
            date=(2022, 5, 17, 6, 0, 0), discipline=0):
    """A GRIB-2 message written from the section layout (WMO FM 92, edition 2): identification, one grid (template 3.0
    or 3.40), then sections 4 - 7 once per field.  `fields`: dicts with values, category, number, and optionally
    surface (type, value), step, nbits, decimal, bitmap, pdt."""
    y, mo, d, h, mi, sec_ = date
    s1 = (98).to_bytes(2, "big") + (0).to_bytes(2, "big") + bytes([27, 0, 1]) + y.to_bytes(2, "big") + \
        bytes([mo, d, h, mi, sec_, 0, 1])
    n_points = int(np.sum(pl)) if pl is not None else ni * nj
    g = bytearray(58)                                    # template octets 15 .. 72
    g[0] = 6
    g[16:20] = (0xFFFFFFFF if pl is not None else ni).to_bytes(4, "big")
    g[20:24] = nj.to_bytes(4, "big")
    g[24:28], g[28:32] = (0).to_bytes(4, "big"), (0xFFFFFFFF).to_bytes(4, "big")
    micro = lambda v: sm(int(round(v * 1e6)), 4)         # noqa: E731
    g[32:36], g[36:40] = micro(la1), micro(lo1)
    g[40] = 48
    g[41:45], g[45:49] = micro(la2), micro(lo2)
    g[49:53] = (0xFFFFFFFF).to_bytes(4, "big")
    g[53:57] = int(n_or_dj).to_bytes(4, "big")
    g[57] = scan
    opt = b"" if pl is None else b"".join(int(p).to_bytes(2, "big") for p in pl)
    s3 = bytes([0]) + n_points.to_bytes(4, "big") + bytes([2 if pl is not None else 0, 1 if pl is not None else 0]) + \
        template.to_bytes(2, "big") + bytes(g) + opt
    body = section(1, s1) + section(3, s3)
    for f in fields:
        vals = np.asarray(f["values"], dtype=np.float64).ravel()
        bitmap = f.get("bitmap")
        present = vals if bitmap is None else vals[np.asarray(bitmap, bool).ravel()]
        nbits, decimal = f.get("nbits", 16), f.get("decimal", 0)
        scaled = present * 10.0 ** decimal
        ref = float(np.float32(np.floor(scaled.min())))
        span = scaled.max() - ref
        e = 0 if span == 0 else int(np.ceil(np.log2(span / ((1 << nbits) - 1))))
        x = np.round((scaled - ref) / 2.0 ** e).astype(np.uint64)
        bits = ((x[:, None] >> np.arange(nbits - 1, -1, -1, dtype=np.uint64)) & np.uint64(1)).astype(np.uint8).ravel()
        data = np.packbits(bits).tobytes()
        stype, svalue = f.get("surface", (1, 0))
        s4 = (0).to_bytes(2, "big") + int(f.get("pdt", 0)).to_bytes(2, "big") + bytes([f["category"], f["number"], 2, 0, 0]) + \
            (0).to_bytes(2, "big") + bytes([0, 1]) + int(f.get("step", 0)).to_bytes(4, "big") + \
            bytes([stype, 0]) + sm(int(svalue), 4) + bytes([255, 255]) + (0xFFFFFFFF).to_bytes(4, "big")
        s5 = present.size.to_bytes(4, "big") + (0).to_bytes(2, "big") + np.array(ref, dtype=">f4").tobytes() + \
            sm(e, 2) + sm(decimal, 2) + bytes([nbits, 0])
        s6 = bytes([255]) if bitmap is None else bytes([0]) + np.packbits(np.asarray(bitmap, np.uint8).ravel()).tobytes()
        body += section(4, s4) + section(5, s5) + section(6, s6) + section(7, data)
    body += b"7777"
    return b"GRIB" + bytes([0, 0, discipline, 2]) + (16 + len(body)).to_bytes(8, "big") + body


def test_edition_2_round_trip_lonlat_levels_bitmap_and_several_fields_per_message(tmp_path, rng):
    ni, nj = 24, 13
    lat, lon = np.linspace(90, -90, nj), np.arange(ni) * 15.0
    grid = dict(template=0, ni=ni, nj=nj, la1=90.0, lo1=0.0, la2=-90.0, lo2=345.0, n_or_dj=15000000)
    t = {lev: 220.0 + 40.0 * np.cos(np.radians(lat))[:, None] + rng.standard_normal((nj, ni)) + lev / 100.0
         for lev in (85000, 50000)}
    t2m = 280.0 + rng.standard_normal((nj, ni))
    sst = 290.0 + rng.standard_normal((nj, ni))
    sea = rng.random((nj, ni)) > 0.4
    msg_a = encode2([dict(values=t[85000], category=0, number=0, surface=(100, 85000), nbits=20),
                     dict(values=t[50000], category=0, number=0, surface=(100, 50000), nbits=20),
                     dict(values=t2m, category=0, number=0, surface=(103, 2), nbits=12, decimal=1)], **grid)
    msg_b = encode2([dict(values=sst, category=3, number=0, bitmap=sea, nbits=14)], discipline=10, **grid)
    path = tmp_path / "mixed.grib2"
    path.write_bytes(msg_a + msg_b)
    ds = open_dataset(str(path))
    assert ds.attrs["GRIB_edition"] == 2 and set(ds.data_vars) == {"t", "t2m", "sst"}
    assert ds["t"].dims == ("isobaricInhPa", "latitude", "longitude")
    assert ds["t"].coords["isobaricInhPa"].values.tolist() == [500.0, 850.0]                 # Pa -> hPa, sorted
    np.testing.assert_allclose(ds["t"].values[1], t[85000], atol=50.0 / (1 << 20) * 2 + 4e-5)
    np.testing.assert_allclose(ds["t"].values[0], t[50000], atol=50.0 / (1 << 20) * 2 + 4e-5)
    np.testing.assert_allclose(ds["t2m"].values, t2m, atol=0.06)                              # one decimal
    assert np.array_equal(np.isnan(ds["sst"].values), ~sea)
    np.testing.assert_allclose(ds["sst"].values[sea], sst[sea], atol=8.0 / (1 << 14) + 4e-5)
    np.testing.assert_allclose(ds.coords["latitude"].values, lat)
    np.testing.assert_allclose(ds.coords["longitude"].values, lon)


def test_edition_2_gaussian_regular_and_reduced_and_times(tmp_path, rng):
    n = 8
    lat = griblite.gaussian_latitudes(n)
    pl = np.array([8, 12, 16, 20, 24, 28, 32, 36, 36, 32, 28, 24, 20, 16, 12, 8])
    red = [rng.random(pl.sum()) for _ in range(3)]
    msgs = [encode2([dict(values=red[k], category=0, number=0, discipline=2, step=6 * k)], template=40, nj=16,
                    la1=lat[0], lo1=0.0, la2=lat[-1], lo2=348.75, n_or_dj=n, pl=pl, discipline=2) for k in range(3)]
    p = tmp_path / "lsm_reduced.grib2"
    p.write_bytes(b"".join(msgs))
    ds = open_dataset(str(p))
    assert ds["lsm"].dims == ("time", "values") and ds["lsm"].shape == (3, pl.sum())
    assert np.diff(ds["lsm"].coords["time"].values).tolist() == [21600.0, 21600.0]          # steps of 6 h
    np.testing.assert_allclose(ds["lsm"].values, np.stack(red), atol=2.0 / (1 << 16))
    assert np.array_equal(ds.coords["latitude"].values, np.repeat(lat, pl))
    from smmregrid_amd.gridmeta import GridInspector
    assert GridInspector(ds).get_gridtype()[0].kind == "GaussianReduced"
    reg = rng.random((16, 32))
    p2 = tmp_path / "regular_gg.grib2"
    p2.write_bytes(encode2([dict(values=reg[::-1], category=0, number=0)], template=40, ni=32, nj=16, la1=lat[-1], lo1=0.0,
                           la2=lat[0], lo2=348.75, n_or_dj=n, scan=0x40, discipline=2))
    ds2 = open_dataset(str(p2))
    assert ds2["lsm"].attrs["GRIB_gridType"] == "regular_gg"
    np.testing.assert_allclose(ds2.coords["latitude"].values, lat[::-1], atol=1e-6)
    np.testing.assert_allclose(ds2["lsm"].values, reg[::-1], atol=2.0 / (1 << 16))


def test_edition_2_says_what_it_does_not_decode(tmp_path):
    base = dict(template=0, ni=4, nj=3, la1=10.0, lo1=0.0, la2=-10.0, lo2=30.0, n_or_dj=10000000)
    good = encode2([dict(values=np.arange(12.0).reshape(3, 4), category=0, number=0)], **base)
    assert open_dataset_bytes(tmp_path, good)["t"].shape == (3, 4)
    sec3 = good.index(section(3, b"")[4:5], 16 + 21)                  # first byte "3" after section 1
    rot = bytearray(good)
    rot[16 + 21 + 12:16 + 21 + 14] = (1).to_bytes(2, "big")           # template 3.1: rotated lon/lat
    with pytest.raises(griblite.GribUnsupported, match="template 3.1"):
        open_dataset_bytes(tmp_path, bytes(rot))
    ccsds = bytearray(good)
    s5 = good.index(b"\x00\x00\x00\x15\x05")                          # section 5: length 21, number 5
    ccsds[s5 + 9:s5 + 11] = (42).to_bytes(2, "big")
    with pytest.raises(griblite.GribUnsupported, match="CCSDS"):
        open_dataset_bytes(tmp_path, bytes(ccsds))
    with pytest.raises(ValueError):
        open_dataset_bytes(tmp_path, good[:-9])
    assert sec3 > 0


def open_dataset_bytes(tmp_path, raw):
    p = tmp_path / "x.grib2"
    p.write_bytes(raw)
    return open_dataset(str(p))
