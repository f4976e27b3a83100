"""Native first-order conservative weights where one side is HEALPix (gridgen.sampled_conservative_weights):
overlap areas from the pixels' equal-area nested sub-pixels.  CPU only (pure numpy)."""
import numpy as np
import pytest

from smmregrid_amd import gridgen
from smmregrid_amd.gridgen import DEG


def _dense(w):
    S, D = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
    m = np.zeros((D, S))
    np.add.at(m, (w["dst_address"].values - 1, w["src_address"].values - 1), w["remap_matrix"].values[:, 0])
    return m


def _cell_areas(grid):
    return (np.diff(np.sin(grid.lat_b * DEG))[:, None] * (np.diff(grid.lon_b) * DEG)[None, :]).ravel()


def test_ring_index_inverts_the_ring_centres():
    for nside in (1, 2, 8, 32):
        lon, lat = gridgen.healpix_centers(nside, nested=False)
        assert np.array_equal(gridgen.healpix_ring_index(nside, lon, lat), np.arange(12 * nside * nside))
        nlon, nlat = gridgen.healpix_centers(nside, nested=True)
        perm = gridgen.healpix_ring_index(nside, nlon, nlat)
        assert np.array_equal(np.sort(perm), np.arange(12 * nside * nside))
        assert np.allclose(lon[perm], nlon) and np.allclose(lat[perm], nlat)


@pytest.mark.parametrize("src,dst", [("r96x48", "hp8_nested"), ("r180x90", "hp16"), ("F32", "hp8_ring")])
def test_regular_to_healpix_conserves(src, dst):
    """Rows sum to 1 (a constant stays a constant), dst_grid_frac is 1 without a mask, and the area integral of a
    smooth field is kept to the accuracy of the sampled overlap areas."""
    w = gridgen.generate_weights(src, dst, method="con")
    g = gridgen.parse_grid(src)
    m = _dense(w)
    assert np.allclose(m.sum(axis=1), 1.0, rtol=0, atol=1e-12) and (m >= 0).all()
    assert np.allclose(w["dst_grid_frac"].values, 1.0, atol=1e-12)
    lon2, lat2 = g.centers()
    x = 280.0 + 30.0 * np.cos(lat2 * DEG) * np.sin(2 * lon2 * DEG) + 10.0 * np.sin(lat2 * DEG)
    y = m @ x
    pix = 4 * np.pi / y.size
    assert abs((y * pix).sum() - (x * _cell_areas(g)).sum()) / abs((x * _cell_areas(g)).sum()) < 2e-3
    assert x.min() - 1e-9 <= y.min() and y.max() <= x.max() + 1e-9          # convex combinations
    # links are stored sorted by (dst, src), 1-based
    d, s = w["dst_address"].values, w["src_address"].values
    assert d.min() >= 1 and s.min() >= 1 and (np.diff(d.astype(np.int64) * (g.size + 1) + s) > 0).all()


def test_ring_and_nested_targets_hold_the_same_weights():
    a = _dense(gridgen.generate_weights("r72x36", "hp4_nested", method="con"))
    b = _dense(gridgen.generate_weights("r72x36", "hp4_ring", method="con"))
    nlon, nlat = gridgen.healpix_centers(4, nested=True)
    perm = gridgen.healpix_ring_index(4, nlon, nlat)          # nested pixel i is ring pixel perm[i]
    assert np.array_equal(b[perm], a)


def test_healpix_to_regular_conserves_and_converges():
    """HEALPix source: every target cell averages the pixels under it.  The sampled area of a lon/lat cell approaches
    its true area as the sub-pixels get finer."""
    errs = []
    for samples in (16, 256):
        w = gridgen.sampled_conservative_weights("hp16_nested", "r72x36", samples=samples)
        m = _dense(w)
        assert np.allclose(m.sum(axis=1), 1.0, atol=1e-12)
        # column sums x destination areas = the source pixel's area (conservation), to sampling accuracy
        dst = gridgen.parse_grid("r72x36")
        col = (m * _cell_areas(dst)[:, None]).sum(axis=0)
        errs.append(np.abs(col / (4 * np.pi / col.size) - 1.0).max())
    assert errs[1] < errs[0] and errs[1] < 0.08
    lon, lat = gridgen.healpix_centers(16, nested=True)
    x = 5.0 + np.sin(lat * DEG) + 0.5 * np.cos(lon * DEG) * np.cos(lat * DEG)
    y = m @ x
    assert abs((y * _cell_areas(dst)).sum() - (x * 4 * np.pi / x.size).sum()) / (x * 4 * np.pi / x.size).sum() < 2e-3


def test_masked_source_cells_give_fractions_and_no_links():
    """A masked band of the source: destination pixels wholly inside it get no link and frac 0, pixels across its
    edge a fraction in (0, 1) and rows that still sum to 1 (fracarea) or to the fraction (destarea)."""
    g = gridgen.parse_grid("r96x48")
    mask = np.ones((48, 96), np.int32)
    mask[18:30, 10:50] = 0
    w = gridgen.generate_weights(g, "hp8_nested", method="con", src_mask=mask.ravel())
    frac = w["dst_grid_frac"].values
    rows = np.bincount(w["dst_address"].values - 1, weights=w["remap_matrix"].values[:, 0], minlength=frac.size)
    assert (frac == 0).any() and ((frac > 0) & (frac < 1)).any() and (frac <= 1).all()
    assert np.allclose(rows[frac > 0], 1.0, atol=1e-12) and (rows[frac == 0] == 0).all()
    assert (mask.ravel()[w["src_address"].values - 1] == 1).all()
    wd = gridgen.generate_weights(g, "hp8_nested", method="con", src_mask=mask.ravel(), norm="destarea")
    rows_d = np.bincount(wd["dst_address"].values - 1, weights=wd["remap_matrix"].values[:, 0], minlength=frac.size)
    assert np.allclose(rows_d, frac, atol=1e-12)
    # the unmasked share of pixel area: band area / sphere within the sampling error
    band = (np.sin(g.lat_b[30] * DEG) - np.sin(g.lat_b[18] * DEG)) * (g.lon_b[50] - g.lon_b[10]) * DEG
    assert abs((1 - frac).sum() * (4 * np.pi / frac.size) - band) / band < 0.02


def test_healpix_to_healpix_uses_the_nested_hierarchy():
    """Coarsening averages the 4^k children exactly, refining copies the parent; ring order on either side; a masked
    child lowers dst_grid_frac by 1 / 4^k."""
    fine = gridgen.parse_grid("hp8_nested")
    x = 2.0 + np.sin(fine.lat * DEG) * np.cos(fine.lon * DEG)
    m = _dense(gridgen.generate_weights("hp8_nested", "hp2_nested", method="con"))
    assert np.allclose(m @ x, x.reshape(48, 16).mean(axis=1), atol=1e-14)
    up = _dense(gridgen.generate_weights("hp2_nested", "hp8_nested", method="con"))
    assert np.array_equal(up, np.kron(np.eye(48), np.ones((16, 1))))
    # ring order on both sides: the same averages, renumbered
    ring_f, ring_c = gridgen.parse_grid("hp8_ring"), gridgen.parse_grid("hp2_ring")
    mr = _dense(gridgen.generate_weights(ring_f, ring_c, method="con"))
    pf = gridgen.healpix_ring_index(8, fine.lon, fine.lat)                 # nested pixel i is ring pixel pf[i]
    coarse = gridgen.parse_grid("hp2_nested")
    pc = gridgen.healpix_ring_index(2, coarse.lon, coarse.lat)
    xr = np.empty_like(x)
    xr[pf] = x
    assert np.allclose((mr @ xr)[pc], m @ x, atol=1e-14)
    mask = np.ones(768, np.int32)
    mask[:4] = 0                                                           # four children of coarse pixel 0
    w = gridgen.generate_weights("hp8_nested", "hp2_nested", method="con", src_mask=mask)
    assert np.isclose(w["dst_grid_frac"].values[0], 12 / 16) and np.allclose(w["dst_grid_frac"].values[1:], 1.0)
    with pytest.raises(ValueError):
        gridgen.generate_weights("r36x18", gridgen.Grid("points", fine.lon, fine.lat, cdo_type="unstructured"), method="con")


@pytest.mark.parametrize("spec", ["hp8_nested", "hp8_ring"])
def test_bilinear_from_a_healpix_source(spec):
    """basic_test.py:82-93 ('hp32' -> r360x180, bil) and :14-29 (hp1 source, bil): the ring-wise 4-point scheme.
    Weights are >= 0 and sum to 1, a point on a pixel centre takes that pixel's value, a smooth field is reproduced
    with an error that falls with the resolution, ring and nested order agree."""
    src = gridgen.parse_grid(spec)
    w = gridgen.generate_weights(spec, "r72x36", method="bil")
    assert w.sizes["src_grid_size"] == 768 and w.sizes["dst_grid_size"] == 72 * 36 and w.sizes["num_links"] == 4 * 72 * 36
    m = _dense(w)
    assert np.allclose(m.sum(axis=1), 1.0, atol=1e-13) and m.min() >= 0.0
    ident = _dense(gridgen.generate_weights(spec, gridgen.Grid("points", src.lon, src.lat, cdo_type="unstructured"), method="bil"))
    assert np.allclose(ident, np.eye(768), atol=1e-9)

    def f(lo, la):
        return 3.0 + np.sin(la * DEG) + 0.5 * np.cos(lo * DEG) * np.cos(la * DEG)
    dl, dla = gridgen.parse_grid("r72x36").centers()
    err8 = np.abs(m @ f(src.lon, src.lat) - f(dl, dla)).max()
    fine = gridgen.parse_grid("hp32" + spec[3:])
    err32 = np.abs(_dense(gridgen.generate_weights(fine, "r72x36", method="bil")) @ f(fine.lon, fine.lat) - f(dl, dla)).max()
    assert err32 < err8 / 4 and err32 < 3e-3
    other = gridgen.parse_grid("hp8_ring" if spec.endswith("nested") else "hp8_nested")
    y_other = _dense(gridgen.generate_weights(other, "r72x36", method="bil")) @ f(other.lon, other.lat)
    assert np.allclose(m @ f(src.lon, src.lat), y_other, atol=1e-12)


def test_bilinear_from_healpix_drops_masked_pixels():
    src = gridgen.parse_grid("hp8_nested")
    mask = np.ones(768, np.int32)
    mask[src.lat > 60.0] = 0
    w = gridgen.generate_weights(src, "r36x18", method="bil", src_mask=mask)
    assert (mask[w["src_address"].values - 1] == 1).all()
    rows = np.bincount(w["dst_address"].values - 1, weights=w["remap_matrix"].values[:, 0], minlength=36 * 18)
    assert set(np.round(rows, 12)) <= {0.0, 1.0} and (rows == 0).any() and (rows == 1).any()


def test_reference_generation_sizes_from_grid_names():
    """basic_test.py:82-93: weights between CDO grid names carry the grid sizes (the hp32 source case needs the
    HEALPix bilinear scheme)."""
    for s, t, ns, nd in (("r180x90", "r360x180", 180 * 90, 360 * 180), ("F128", "r180x90", 256 * 512, 180 * 90),
                         ("hp32", "r360x180", 12288, 360 * 180)):
        w = gridgen.generate_weights(s, t, method="bil")
        assert w.sizes["src_grid_size"] == ns and w.sizes["dst_grid_size"] == nd


def test_distance_weighted_average_of_four_neighbours():
    """`dis` (basic_test.py:61-68 uses it): four nearest source centres, weights 1 / great-circle distance normalised,
    a coinciding centre takes all; masked cells are not candidates; north-to-south files keep their cell order."""
    w = gridgen.generate_weights("r96x48", "r36x18", method="dis")
    assert w.sizes["num_links"] == 4 * 36 * 18 and w.attrs["map_method"].startswith("Distance")
    m = _dense(w)
    assert np.allclose(m.sum(axis=1), 1.0, atol=1e-13) and m.min() >= 0 and ((m > 0).sum(axis=1) == 4).all()
    g = gridgen.parse_grid("r96x48")
    lon2, lat2 = g.centers()
    same = _dense(gridgen.generate_weights(g, gridgen.Grid("points", lon2[100:160], lat2[100:160], cdo_type="unstructured"), method="dis"))
    assert np.array_equal(same, np.eye(96 * 48)[100:160])
    # the weights of one target point against the definition
    d = gridgen.parse_grid("r36x18")
    dl, dla = d.centers()
    k = 300
    v = lambda lo, la: np.stack([np.cos(la * DEG) * np.cos(lo * DEG), np.cos(la * DEG) * np.sin(lo * DEG), np.sin(la * DEG)], -1)
    ang = np.arccos(np.clip(v(lon2, lat2) @ v(dl[k], dla[k]), -1, 1))
    near = np.argsort(ang)[:4]
    ref = (1 / ang[near]) / (1 / ang[near]).sum()
    assert np.allclose(np.sort(m[k][m[k] > 0]), np.sort(ref), rtol=1e-9)
    mask = np.ones(96 * 48, np.int32)
    mask[near] = 0
    wm = gridgen.generate_weights(g, d, method="dis", src_mask=mask)
    assert (mask[wm["src_address"].values - 1] == 1).all()
    # a source stored north to south: the same numbers, cells renumbered
    desc = gridgen.regular_grid_from_centers(g.lon, g.lat[::-1])
    md = _dense(gridgen.generate_weights(desc, d, method="dis"))
    assert np.allclose(md.reshape(-1, 48, 96)[:, ::-1].reshape(-1, 96 * 48), m, atol=1e-14)


def test_largest_area_fraction_takes_the_heaviest_conservative_link():
    """`laf`: one link of weight 1 per destination cell, to the unmasked source cell with the largest overlap
    (the heaviest link of the conservative weights of the same pair; the lowest address among equals)."""
    w = gridgen.generate_weights("r96x48", "r36x18", method="laf")
    con = _dense(gridgen.generate_weights("r96x48", "r36x18", method="con"))
    assert w.sizes["num_links"] == 36 * 18 and set(w["remap_matrix"].values[:, 0]) == {1.0}
    assert np.array_equal(w["src_address"].values - 1, con.argmax(axis=1))
    g = gridgen.parse_grid("r96x48")
    a = w["src_address"].values - 1
    desc = gridgen.regular_grid_from_centers(g.lon, g.lat[::-1])           # the same grid stored north to south
    b = gridgen.generate_weights(desc, "r36x18", method="laf")["src_address"].values - 1
    assert np.array_equal((47 - b // 96) * 96 + b % 96, a)
    mask = np.ones(96 * 48, np.int32)
    mask[a[:100]] = 0
    wm = gridgen.generate_weights("r96x48", "r36x18", method="laf", src_mask=mask)
    assert (mask[wm["src_address"].values - 1] == 1).all() and wm.sizes["num_links"] == 36 * 18
    hp = gridgen.generate_weights("r96x48", "hp4_nested", method="laf")    # through the sampled overlaps
    assert hp.sizes["num_links"] == 192 and set(hp["remap_matrix"].values[:, 0]) == {1.0}


def test_bicubic_weights_carry_four_columns_and_column_zero_is_the_value_basis():
    """`bic` (basic_test.py:31-39 uses it): SCRIP's four weights per corner; the reference applies column 0 only, the
    cubic Hermite value basis -- a partition of unity that reproduces the nodes and, between them, the smoothstep of
    the bilinear fractions."""
    w = gridgen.generate_weights("r96x48", "r36x18", method="bic")
    rm = w["remap_matrix"].values
    assert rm.shape == (4 * 36 * 18, 4) and w.sizes["num_wgts"] == 4
    rows = np.bincount(w["dst_address"].values - 1, weights=rm[:, 0], minlength=36 * 18)
    assert np.allclose(rows, 1.0, atol=1e-13) and rm[:, 0].min() >= 0.0
    bil = gridgen.generate_weights("r96x48", "r36x18", method="bil")
    assert np.array_equal(bil["src_address"].values, w["src_address"].values)
    # per destination the bicubic value weights are the products of the smoothstepped 1-D fractions of the bilinear ones
    wb = bil["remap_matrix"].values[:, 0].reshape(-1, 4)
    fx = wb[:, 1] + wb[:, 3]          # links sorted by source address: (j0,i0), (j0,i1), (j1,i0), (j1,i1) unless the row wraps
    fy = wb[:, 2] + wb[:, 3]
    sm = lambda t: 3 * t ** 2 - 2 * t ** 3
    ref = np.stack([(1 - sm(fx)) * (1 - sm(fy)), sm(fx) * (1 - sm(fy)), (1 - sm(fx)) * sm(fy), sm(fx) * sm(fy)], axis=1)
    inner = (w["src_address"].values.reshape(-1, 4)[:, 1] - w["src_address"].values.reshape(-1, 4)[:, 0]) == 1
    assert inner.sum() > 500 and np.allclose(rm[:, 0].reshape(-1, 4)[inner], ref[inner], atol=1e-12)
    g = gridgen.parse_grid("r96x48")
    lon2, lat2 = g.centers()
    node = gridgen.generate_weights(g, gridgen.Grid("points", lon2[200:260], lat2[200:260], cdo_type="unstructured"), method="bic")
    m = _dense(node)
    assert np.allclose(m, np.eye(96 * 48)[200:260], atol=1e-12)
    mask = np.ones(96 * 48, np.int32)
    mask[1000:1200] = 0
    wm = gridgen.generate_weights("r96x48", "r36x18", method="bic", src_mask=mask)
    assert (mask[wm["src_address"].values - 1] == 1).all()
    rows = np.bincount(wm["dst_address"].values - 1, weights=wm["remap_matrix"].values[:, 0], minlength=36 * 18)
    assert np.allclose(rows[rows > 0], 1.0, atol=1e-13)


def test_second_order_conservative_weights_follow_their_definition():
    """`con2` (SCRIP, Jones 1999 eqs. 4 - 6), three columns: column 0 is the first-order weight (what the reference
    applies); with the latitude gradient column a field linear in latitude is remapped exactly; columns 1 and 2 equal
    a brute-force quadrature of int (lat - lat_n) dA and int cos(lat) (lon - lon_n) dA over the overlap -- also for
    the source cells that straddle longitude 0."""
    src, dst = gridgen.parse_grid("r96x48"), gridgen.parse_grid("r36x18")
    w = gridgen.generate_weights(src, dst, method="con2")
    first = gridgen.generate_weights(src, dst, method="con")
    rm = w["remap_matrix"].values
    assert rm.shape[1] == 3 and np.array_equal(rm[:, 0], first["remap_matrix"].values[:, 0])
    assert np.array_equal(w["src_address"].values, first["src_address"].values)
    s, d = w["src_address"].values - 1, w["dst_address"].values - 1
    sb, db = src.lat_b * DEG, dst.lat_b * DEG
    cen = lambda b: ((b[1:] * np.sin(b[1:]) + np.cos(b[1:])) - (b[:-1] * np.sin(b[:-1]) + np.cos(b[:-1]))) / (np.sin(b[1:]) - np.sin(b[:-1]))
    f = 2.0 + 3.0 * np.repeat(cen(sb), 96)                     # cell means of 2 + 3 lat
    y2 = np.bincount(d, weights=rm[:, 0] * f[s] + rm[:, 1] * 3.0, minlength=648)
    y1 = np.bincount(d, weights=rm[:, 0] * f[s], minlength=648)
    exact = 2.0 + 3.0 * np.repeat(cen(db), 36)
    assert np.abs(y2 - exact).max() < 1e-13 < 1e-3 < np.abs(y1 - exact).max()
    # brute force of the definition on a sample of links (incl. source column 0, centred on longitude 0)
    rng = np.random.default_rng(5)
    pick = np.concatenate([rng.choice(s.size, 40, replace=False), np.flatnonzero(s % 96 == 0)[:10]])
    covered = np.bincount(d, weights=0 * rm[:, 0] + 1, minlength=648)         # every cell is covered (no mask)
    assert covered.min() > 0
    dst_area = np.repeat(np.diff(np.sin(db)), 36) * (10.0 * DEG)
    for k in pick:
        js, i_s, jd, i_d = s[k] // 96, s[k] % 96, d[k] // 36, d[k] % 36
        lat_n, lon_n = cen(sb)[js], src.lon[i_s]
        t = np.linspace(max(sb[js], db[jd]), min(sb[js + 1], db[jd + 1]), 2001)
        lo1 = ((dst.lon_b[i_d] - lon_n + 180) % 360) - 180
        lo2 = lo1 + 10.0
        a, b = max(lo1, -1.875), min(lo2, 1.875)
        if b <= a:
            a, b = max(lo1 + 360, -1.875), min(lo2 + 360, 1.875)
        p = np.linspace(a, b, 401) * DEG
        tm, pm = 0.5 * (t[1:] + t[:-1]), 0.5 * (p[1:] + p[:-1])
        dA = np.outer(np.cos(tm) * np.diff(t), np.diff(p))
        w1 = dA.sum()
        w2 = ((tm - lat_n)[:, None] * dA).sum()
        w3 = (np.cos(tm)[:, None] * pm[None, :] * dA).sum()
        assert np.isclose(w1 / dst_area[d[k]], rm[k, 0], rtol=1e-6)
        assert np.isclose(w2 / dst_area[d[k]], rm[k, 1], rtol=1e-4, atol=1e-8)
        assert np.isclose(w3 / dst_area[d[k]], rm[k, 2], rtol=1e-4, atol=1e-8)


def test_healpix_field_with_coordinates_in_radians_is_recognised():
    """tests/data/tas-healpix2.nc of the reference (identity2d_test.py:14-18; fixture tests/golden/tas_healpix2.npz):
    a nested nside-32 field whose lon(pix) / lat(pix) are in radians.  CdoGenerate converts by the `units` attribute
    and recognises the pixel centres as a HEALPix grid, so conservative weights are available for it."""
    import os
    from smmregrid_amd import CdoGenerate, DataArray
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "tas_healpix2.npz"))
    assert str(z["units"]) == "radian"
    coords = {"lat": DataArray(z["lat"], dims=("pix",), attrs={"units": "radian"}),
              "lon": DataArray(z["lon"], dims=("pix",), attrs={"units": "radian"})}
    field = DataArray(z["tas"], dims=("time", "pix"), coords=coords, name="tas")
    g = CdoGenerate._grid_of(field)
    assert g.cdo_type == "healpix" and g.nside == 32 and g.nested is True and g.size == 12288
    ref = gridgen.parse_grid("hp32_nested")
    assert np.allclose(g.lon, ref.lon) and np.allclose(g.lat, ref.lat)
    # ring order and non-HEALPix lists
    rlon, rlat = gridgen.healpix_centers(8, nested=False)
    assert gridgen.healpix_grid_of_centers(rlon, rlat).nested is False
    assert gridgen.healpix_grid_of_centers(rlon[::-1], rlat[::-1]) is None
    assert gridgen.healpix_grid_of_centers(rlon[:100], rlat[:100]) is None
    # degrees stay degrees
    plain = DataArray(z["tas"], dims=("time", "pix"),
                      coords={"lat": DataArray(np.degrees(z["lat"]), dims=("pix",), attrs={"units": "degrees_north"}),
                              "lon": DataArray(np.degrees(z["lon"]), dims=("pix",), attrs={"units": "degrees_east"})}, name="tas")
    assert CdoGenerate._grid_of(plain).nside == 32
    w = gridgen.generate_weights(g, "r72x36", method="con")
    m = _dense(w)
    y = m @ z["tas"][0].astype(np.float64)
    assert np.allclose(m.sum(axis=1), 1.0) and 200.0 < y.min() and y.max() < 330.0
    mean_src = z["tas"][0].astype(np.float64).mean()                       # equal-area pixels
    dst = gridgen.parse_grid("r72x36")
    assert abs((y * _cell_areas(dst)).sum() / (4 * np.pi) - mean_src) < 0.05


def test_cell_edges_come_from_the_files_bounds_variables():
    """A lon/lat file that carries lat_bnds / lon_bnds (CF `bounds` attribute; tests/data/tas-ecearth.nc of the reference:
    a Gaussian grid whose first cell reaches the pole) gives its cell edges to the native generator, as it does to CDO;
    without them -- or when they are not contiguous -- the edges are the mid-points.  Either latitude direction."""
    from smmregrid_amd import CdoGenerate, DataArray, Dataset
    g = gridgen.gaussian_grid(16)
    lat_b = g.lat_b.copy()
    lat_b[3] += 0.05                                            # a file's edges need not be mid-points
    lat_bnds = np.stack([lat_b[:-1], lat_b[1:]], axis=1)
    lon_bnds = np.stack([g.lon_b[:-1], g.lon_b[1:]], axis=1)
    x = np.ones((2, g.lat.size, g.lon.size))

    def dataset(lat, lat_bnds):
        coords = {"time": np.arange(2), "lat": DataArray(lat, dims=("lat",), attrs={"units": "degrees_north", "bounds": "lat_bnds"}),
                  "lon": DataArray(g.lon, dims=("lon",), attrs={"units": "degrees_east", "bounds": "lon_bnds"})}
        ds = Dataset({"tas": DataArray(x, dims=("time", "lat", "lon"), coords=coords, name="tas")})
        ds["lat_bnds"] = (("lat", "bnds"), lat_bnds)
        ds["lon_bnds"] = (("lon", "bnds"), lon_bnds)
        return ds

    ds = dataset(g.lat, lat_bnds)
    got = CdoGenerate._grid_of(ds)
    assert np.allclose(got.lat_b, lat_b) and np.allclose(got.lon_b, g.lon_b) and not got.lat_descending
    alone = CdoGenerate._grid_of(ds["tas"])                     # no bounds in reach: mid-points
    assert not np.allclose(alone.lat_b, lat_b) and np.isclose(alone.lat_b[1], 0.5 * (g.lat[0] + g.lat[1]))
    desc = CdoGenerate._grid_of(dataset(g.lat[::-1].copy(), lat_bnds[::-1, ::-1].copy()))
    assert desc.lat_descending and np.allclose(desc.lat_b, lat_b)
    broken = lat_bnds.copy()
    broken[5, 1] += 0.3                                         # a gap between two cells: not usable
    assert np.allclose(CdoGenerate._grid_of(dataset(g.lat, broken)).lat_b, alone.lat_b)
    # the conservative weights notice: the areas of the moved cells change, rows still sum to one
    w = gridgen.generate_weights(got, "r18x9", method="con")
    w0 = gridgen.generate_weights(alone, "r18x9", method="con")
    assert not np.array_equal(w["remap_matrix"].values, w0["remap_matrix"].values)
    assert np.allclose(np.bincount(w["dst_address"].values - 1, weights=w["remap_matrix"].values[:, 0]), 1.0)


def test_areas_of_a_healpix_field_with_coordinates():
    """areas_test.py:17-31, :34-46 with tas-healpix2.nc as source and as target: 12288 equal cells that add up to
    the Earth's surface (the grid is recognised from its pixel centres in radians)."""
    import os
    from smmregrid_amd import CdoGenerate, DataArray
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "tas_healpix2.npz"))
    coords = {"lat": DataArray(z["lat"], dims=("pix",), attrs={"units": "radian"}),
              "lon": DataArray(z["lon"], dims=("pix",), attrs={"units": "radian"})}
    field = DataArray(z["tas"], dims=("time", "pix"), coords=coords, name="tas")
    earth = 5.101e8
    for gen, kw in ((CdoGenerate(field, "r360x180"), {}), (CdoGenerate("r360x180", field), {"target": True})):
        area = gen.areas(**kw)
        assert area["cell_area"].shape == (12288,)
        assert abs(area["cell_area"].values.sum() / 1e6 - earth) < 0.02 * earth
        assert np.allclose(area["cell_area"].values, area["cell_area"].values[0])


def test_regional_lonlat_sources_do_not_wrap():
    """A regional lon/lat source (the layout of the reference's tests/data/regional.nc: 61 x 90 cells of one degree):
    bilinear / bicubic corners are found by position, not by assuming 360 / nx spacing, points beyond the region take
    its edge column (extrapolation by the nearest edge), nearest neighbour searches the centres; global grids keep the
    equally spaced periodic rule bit for bit."""
    lon, lat = np.arange(0.0, 61.0), np.arange(-19.5, 70.5)
    reg = gridgen.regular_grid_from_centers(lon, lat)
    assert not gridgen._is_cyclic(reg) and gridgen._is_cyclic(gridgen.parse_grid("r180x90")) and gridgen._is_cyclic(gridgen.gaussian_grid(16))
    pts = gridgen.Grid("points", np.array([10.25, 59.9, 75.0, 350.0]), np.array([0.3, 10.0, 10.0, 10.0]), cdo_type="unstructured")
    f = lambda lo, la: 2.0 + 0.1 * lo + 0.05 * la                          # bilinear reproduces a plane inside the region
    lon2, lat2 = reg.centers()
    for method in ("bil", "bic"):
        m = _dense(gridgen.generate_weights(reg, pts, method=method))
        assert np.allclose(m.sum(axis=1), 1.0)
        y = m @ f(lon2, lat2)
        if method == "bil":
            assert np.allclose(y[:2], f(pts.lon[:2], pts.lat[:2]), atol=1e-12)
        assert np.isclose(y[2], f(60.0, 10.0)) and np.isclose(y[3], f(0.0, 10.0))       # clamped to the edge columns
    nn = gridgen.generate_weights(reg, pts, method="nn")
    cols = (nn["src_address"].values - 1) % 61
    assert list(cols) == [10, 60, 60, 0]
    con = gridgen.generate_weights(reg, "r36x18", method="con")            # cells outside the region stay uncovered
    frac = con["dst_grid_frac"].values.reshape(18, 36)
    assert frac[:, 10:].max() == 0.0 and np.allclose(frac[8:15, 1:5], 1.0)


def test_regional_lonlat_grid_and_healpix_conservative():
    """a REGIONAL lon/lat grid against a HEALPix grid, either side.  Sub-pixels outside the
    region used to be clipped into its border cells, so every pixel of the sphere got links (dst_grid_frac == 1
    everywhere) and edge cells piled up dozens of links.  Now only what the region covers is linked."""
    lon, lat = np.arange(0.5, 40.0), np.arange(30.5, 60.0)                 # 0-40 E, 30-60 N, one-degree cells
    reg = gridgen.regular_grid_from_centers(lon, lat)
    region_area = np.radians(40.0) * (np.sin(np.radians(60.0)) - np.sin(np.radians(30.0)))
    hp_lon, hp_lat = gridgen.healpix_centers(8, nested=True)
    pix_area = 4 * np.pi / 768
    # regional -> hp8: only pixels that overlap the region are linked, their fraction is the covered share
    w = gridgen.generate_weights(reg, "hp8", method="con")
    frac = w["dst_grid_frac"].values
    linked = np.unique(w["dst_address"].values - 1)
    assert 8 <= linked.size <= 40 and np.array_equal(np.flatnonzero(frac > 0), linked)
    assert abs((frac * pix_area).sum() - region_area) < 6e-3 * region_area          # the region's area, no more
    inside = (hp_lon > 4) & (hp_lon < 36) & (hp_lat > 36) & (hp_lat < 54)
    assert inside.sum() >= 3 and np.allclose(frac[inside], 1.0, atol=1e-3)          # deep inside: fully covered
    far = (hp_lon > 90) & (hp_lon < 300)
    assert frac[far].max() == 0.0
    rows = np.bincount(w["dst_address"].values - 1, weights=w["remap_matrix"].values[:, 0], minlength=768)
    assert np.allclose(rows[linked], 1.0)
    # hp8 -> regional: every cell of the region is covered by the pixels over it, edge cells hold no pile of links
    w = gridgen.generate_weights("hp8", reg, method="con")
    assert np.allclose(w["dst_grid_frac"].values, 1.0)
    used = np.unique(w["src_address"].values - 1)
    assert used.size <= 45 and not (set(used) & set(np.flatnonzero(far)))           # only pixels over the region
    per_cell = np.bincount(w["dst_address"].values - 1, minlength=reg.size).reshape(lat.size, lon.size)
    assert per_cell.max() <= 6 and per_cell[0, 0] <= 4 and per_cell[-1, -1] <= 4
    # the integral of a smooth field over the region is kept (first order, cells of 1 degree under pixels of 7)
    field = 280.0 + 20.0 * np.cos(np.radians(hp_lat)) * np.cos(np.radians(hp_lon))
    y = np.bincount(w["dst_address"].values - 1, weights=w["remap_matrix"].values[:, 0] * field[w["src_address"].values - 1],
                    minlength=reg.size)
    assert field.min() - 1e-9 <= y.min() and y.max() <= field.max() + 1e-9


def test_fine_regional_target_under_few_samples_has_no_holes():
    """cells of a fine REGIONAL lon/lat target that catch no sub-pixel centre (few `samples`) take the
    pixel holding their own centre, as cells of a global target do: no NaN holes inside the covered area."""
    lon, lat = np.arange(10.05, 14.0, 0.1), np.arange(40.05, 43.0, 0.1)     # 0.1-degree cells under 7-degree pixels
    reg = gridgen.regular_grid_from_centers(lon, lat)
    w = gridgen.sampled_conservative_weights("hp8", reg, samples=16)        # 16 sub-pixels per pixel: ~1.8 degrees each
    per_cell = np.bincount(w["dst_address"].values - 1, minlength=reg.size)
    assert per_cell.min() >= 1 and (w["dst_grid_frac"].values > 0).all()
    rows = np.bincount(w["dst_address"].values - 1, weights=w["remap_matrix"].values[:, 0], minlength=reg.size)
    assert np.allclose(rows, 1.0)
    # the pixel a hole-filling link names is the one that holds the cell's centre
    clon, clat = reg.centers()
    nlon, nlat = gridgen.healpix_centers(8, nested=True)
    ring2nest = np.argsort(gridgen.healpix_ring_index(8, nlon, nlat))
    holder = ring2nest[gridgen.healpix_ring_index(8, clon, clat)]
    single = np.flatnonzero(per_cell == 1)
    order = np.argsort(w["dst_address"].values, kind="stable")
    first_link = order[np.searchsorted(w["dst_address"].values[order], single + 1)]
    assert np.array_equal(w["src_address"].values[first_link] - 1, holder[single])


def test_curvilinear_centres_feed_nn_and_dis():
    """A NEMO-style field (2-D nav_lon / nav_lat on (y, x); the reference's so3d-nemo.nc / onlytos-ipsl.nc layout):
    the native generator takes the cell centres in storage order for `nn` and `dis`, reports (nx, ny) as
    src_grid_dims, and says that the cell-shape methods need regular grids."""
    from smmregrid_amd import CdoGenerate, DataArray
    ny, nx = 20, 30
    j, i = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    nav_lon = (i * 12.0 + j * 1.5) % 360.0
    nav_lat = -80.0 + j * 8.0 + 0.5 * np.sin(i)
    f = DataArray(np.ones((2, ny, nx)), dims=("time", "y", "x"),
                  coords={"nav_lon": DataArray(nav_lon, dims=("y", "x")), "nav_lat": DataArray(nav_lat, dims=("y", "x"))}, name="tos")
    g = CdoGenerate._grid_of(f)
    assert g.kind == "points" and g.cdo_type == "curvilinear" and list(g.dims) == [nx, ny] and g.size == nx * ny
    w = gridgen.generate_weights(g, "r36x18", method="nn")
    assert list(w["src_grid_dims"].values) == [nx, ny] and w.sizes["num_links"] == 36 * 18
    d = gridgen.parse_grid("r36x18")
    dl, dla = d.centers()
    v = lambda lo, la: np.stack([np.cos(la * DEG) * np.cos(lo * DEG), np.cos(la * DEG) * np.sin(lo * DEG), np.sin(la * DEG)], -1)
    best = (v(dl, dla) @ v(nav_lon.ravel(), nav_lat.ravel()).T).argmax(axis=1)
    assert np.array_equal(w["src_address"].values - 1, best)
    assert gridgen.generate_weights(g, "r36x18", method="dis").sizes["num_links"] == 4 * 36 * 18
    with pytest.raises(ValueError, match="regular"):
        gridgen.generate_weights(g, "r36x18", method="con")
