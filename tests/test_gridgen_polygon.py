"""Native conservative weights from POLYGON source cells (unstructured meshes with vertex bounds:
the reference's tests/identity3d_test.py:28-32 runs `con` on temp3d-fesom.nc -> r360x180 through cdo).
Geometry checks against cells whose areas are known, then the reference's own mesh (fixture
tests/golden/temp3d_fesom.npz, made by tests/golden/make_ref_data_fixtures.py)."""
import os

import numpy as np
import pytest

from smmregrid_amd import gridgen
from smmregrid_amd.gridgen import Grid

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def voronoi_mesh(n, seed, pad_to=None):
    """Great-circle polygons that tile the sphere: the Voronoi cells of n random points.  Short polygons are padded by
    repeating their last vertex, as CDO writes unstructured grids."""
    from scipy.spatial import SphericalVoronoi
    rng = np.random.default_rng(seed)
    p = rng.standard_normal((n, 3))
    p /= np.linalg.norm(p, axis=1)[:, None]
    sv = SphericalVoronoi(p, radius=1.0)
    sv.sort_vertices_of_regions()
    V = pad_to or max(len(r) for r in sv.regions)
    lon_v, lat_v = np.empty((n, V)), np.empty((n, V))
    for i, reg in enumerate(sv.regions):
        v = sv.vertices[reg + [reg[-1]] * (V - len(reg))]
        lon_v[i] = np.degrees(np.arctan2(v[:, 1], v[:, 0])) % 360.0
        lat_v[i] = np.degrees(np.arcsin(np.clip(v[:, 2], -1, 1)))
    g = Grid("points", np.degrees(np.arctan2(p[:, 1], p[:, 0])) % 360.0, np.degrees(np.arcsin(p[:, 2])),
             name="voronoi", cdo_type="unstructured")
    g.vertices = (lon_v, lat_v)
    return g, sv.calculate_areas()


def link_areas(w):
    """Overlap area of every link from fracarea weights: w = area / covered(dst), covered = frac * dst area."""
    d = w["dst_address"].values - 1
    return w["remap_matrix"].values[:, 0] * (w["dst_grid_frac"].values * w["dst_grid_area"].values)[d]


def test_a_mesh_that_tiles_the_sphere_covers_every_target_cell_and_keeps_cell_areas():
    src, areas = voronoi_mesh(400, seed=3, pad_to=14)
    w = gridgen.polygon_conservative_weights(src, "r72x36", samples=8)
    assert w.sizes["src_grid_size"] == 400 and w.sizes["dst_grid_size"] == 72 * 36
    np.testing.assert_allclose(w["dst_grid_frac"].values, 1.0, atol=1e-12)          # no gaps between great-circle cells
    d, s = w["dst_address"].values - 1, w["src_address"].values - 1
    rows = np.bincount(d, weights=w["remap_matrix"].values[:, 0], minlength=72 * 36)
    np.testing.assert_allclose(rows, 1.0, atol=1e-12)
    got = np.bincount(s, weights=link_areas(w), minlength=400)                       # area of each cell, re-assembled
    np.testing.assert_allclose(got.sum(), 4 * np.pi, rtol=1e-12)
    rel = np.abs(got - areas) / areas.mean()                                       # sub-cells of 0.625 degrees, cells of 10
    assert rel.max() < 0.04 and rel.mean() < 0.01
    # CDO link order: by destination, then by source
    assert (np.diff(d) >= 0).all() and (np.diff(s)[np.diff(d) == 0] > 0).all()


def test_the_sampling_error_shrinks_with_the_number_of_sub_cells():
    src, areas = voronoi_mesh(150, seed=5)
    err = []
    for m in (2, 4, 8):
        w = gridgen.polygon_conservative_weights(src, "r36x18", samples=m)
        got = np.bincount(w["src_address"].values - 1, weights=link_areas(w), minlength=150)
        err.append(np.abs(got - areas).mean())
    assert err[0] > err[1] > err[2] and err[2] < 0.25 * err[0]


def test_lonlat_boxes_near_the_equator_agree_with_the_exact_generator():
    """A band of lon/lat boxes given as 4-vertex polygons: near the equator parallels are almost great circles, so
    the weights approach the exact (lon, sin lat) ones; outside the band nothing is covered."""
    lon_b, lat_b = np.arange(0.0, 361.0, 6.0), np.arange(-12.0, 13.0, 6.0)
    lo, la = np.meshgrid(0.5 * (lon_b[1:] + lon_b[:-1]), 0.5 * (lat_b[1:] + lat_b[:-1]))
    src = Grid("points", lo.ravel(), la.ravel(), name="band", cdo_type="unstructured")
    w0, w1 = np.meshgrid(lon_b[:-1], lat_b[:-1])
    e0, e1 = np.meshgrid(lon_b[1:], lat_b[1:])
    src.vertices = (np.stack([w0, e0, e0, w0], -1).reshape(-1, 4), np.stack([w1, w1, e1, e1], -1).reshape(-1, 4))
    w = gridgen.polygon_conservative_weights(src, "r90x45", samples=10)
    frac = w["dst_grid_frac"].values.reshape(45, 90)
    dst = gridgen.parse_grid("r90x45")
    inside = (dst.lat_b[1:] <= 12.0) & (dst.lat_b[:-1] >= -12.0)
    outside = (dst.lat_b[:-1] >= 13.0) | (dst.lat_b[1:] <= -13.0)
    np.testing.assert_allclose(frac[inside], 1.0, atol=1e-12)
    assert (frac[outside] == 0.0).all() and not w["dst_address"].values.size == 0
    exact = gridgen.conservative_weights(gridgen.regular_grid_from_centers(lo[0], la[:, 0], lon_b=lon_b, lat_b=lat_b),
                                         "r90x45")
    def dense(ds):
        a = np.zeros((ds.sizes["dst_grid_size"], ds.sizes["src_grid_size"]))
        a[ds["dst_address"].values - 1, ds["src_address"].values - 1] = ds["remap_matrix"].values[:, 0]
        return a
    rows = np.repeat(inside, 90)
    assert np.abs(dense(w)[rows] - dense(exact)[rows]).max() < 0.03     # 1/10-cell sub-cells + parallel vs great circle


def test_masked_cells_drop_their_links_and_lower_the_covered_fraction():
    src, areas = voronoi_mesh(300, seed=11)
    mask = np.ones(300, dtype=np.int32)
    mask[::3] = 0
    full = gridgen.generate_weights(src, "r36x18", method="con")
    part = gridgen.generate_weights(src, "r36x18", method="con", src_mask=mask)
    assert (mask[part["src_address"].values - 1] == 1).all()
    assert part.sizes["num_links"] < full.sizes["num_links"]
    assert (part["dst_grid_frac"].values <= full["dst_grid_frac"].values + 1e-12).all()
    share = (part["dst_grid_frac"].values * part["dst_grid_area"].values).sum() / (4 * np.pi)
    assert share == pytest.approx(areas[mask == 1].sum() / (4 * np.pi), abs=0.01)      # the unmasked cells' area
    assert np.array_equal(part["src_grid_imask"].values, mask)
    # destarea: rows sum to the covered fraction instead of 1
    dest = gridgen.generate_weights(src, "r36x18", method="con", src_mask=mask, norm="destarea")
    rows = np.bincount(dest["dst_address"].values - 1, weights=dest["remap_matrix"].values[:, 0], minlength=36 * 18)
    np.testing.assert_allclose(rows, dest["dst_grid_frac"].values, atol=1e-12)
    with pytest.raises(ValueError, match="cells"):
        gridgen.generate_weights(src, "r36x18", method="con", src_mask=np.ones(299))


def test_grids_without_cells_are_refused():
    src, _ = voronoi_mesh(50, seed=1)
    bare = Grid("points", src.lon, src.lat, name="centres only", cdo_type="unstructured")
    with pytest.raises(ValueError):
        gridgen.polygon_conservative_weights(bare, "r36x18")          # no polygons at all
    with pytest.raises(ValueError, match="no cells"):
        gridgen.polygon_conservative_weights(src, bare)               # polygons -> a bare list of centres
    with pytest.raises(ValueError):
        gridgen.generate_weights(bare, "r36x18", method="con")


def overlap_table(w):
    """(dst, src) -> overlap area, from fracarea weights."""
    d, s = w["dst_address"].values - 1, w["src_address"].values - 1
    return {(int(a), int(b)): float(v) for a, b, v in zip(d, s, link_areas(w))}


def test_polygons_as_the_target_mirror_polygons_as_the_source():
    """lon/lat -> mesh and mesh -> lon/lat count the same overlaps (the samples ride on the lon/lat cells both times),
    so the overlap area of a pair is the same number in both directions; HEALPix on the other side likewise."""
    mesh, areas = voronoi_mesh(90, seed=4)
    for other in ("r36x18", "hp4", "hp4_ring"):
        fwd = gridgen.generate_weights(mesh, other, method="con")                 # mesh -> other
        bwd = gridgen.generate_weights(other, mesh, method="con")                 # other -> mesh
        assert bwd.sizes["dst_grid_size"] == 90 and bwd["dst_grid_dims"].values.tolist() == [90]
        np.testing.assert_allclose(bwd["dst_grid_area"].values, areas, rtol=1e-9)
        np.testing.assert_allclose(bwd["dst_grid_frac"].values, 1.0, atol=1e-12)   # counted on the overlaps' own samples
        # link_areas() scales by the exact cell area; undo to the sampled measure: w * (sampled area of the target cell)
        d, s_ = bwd["dst_address"].values - 1, bwd["src_address"].values - 1
        sampled = np.zeros(90)
        for (dd, ss), v in overlap_table(fwd).items():
            sampled[ss] += v
        b = {(int(ss), int(dd)): float(wv * sampled[dd]) for dd, ss, wv in zip(d, s_, bwd["remap_matrix"].values[:, 0])}
        a = overlap_table(fwd)
        assert a.keys() == b.keys()
        for k, v in a.items():
            assert v == pytest.approx(b[k], rel=1e-9, abs=1e-12), (other, k)
        rows = np.bincount(d, weights=bwd["remap_matrix"].values[:, 0], minlength=90)
        np.testing.assert_allclose(rows, 1.0, atol=1e-12)
        assert np.abs(sampled - areas).max() / areas.mean() < 0.2                # sampling error of the cell areas at the default m
    # ring order is the nested answer, renumbered
    nested = gridgen.generate_weights(mesh, "hp4", method="con")
    ring = gridgen.generate_weights(mesh, "hp4_ring", method="con")
    nlon, nlat = gridgen.healpix_centers(4, nested=True)
    to_ring = gridgen.healpix_ring_index(4, nlon, nlat)
    tn = {(int(to_ring[d]), s): v for (d, s), v in overlap_table(nested).items()}
    tr = overlap_table(ring)
    assert tn.keys() == tr.keys() and all(tn[k] == pytest.approx(tr[k], rel=1e-12) for k in tn)


def test_polygons_on_both_sides():
    """Mesh -> mesh: both are looked up on a fine HEALPix lattice; a constant stays a constant, every target cell is
    covered, and the cell areas re-assembled from the overlaps are the polygons' own."""
    a, area_a = voronoi_mesh(60, seed=21)
    b, area_b = voronoi_mesh(45, seed=22)
    assert gridgen.generate_weights(a, b, method="con").sizes["dst_grid_size"] == 45      # the dispatch finds it
    w = gridgen.polygon_conservative_weights(a, b, samples=8)
    assert w.sizes["src_grid_size"] == 60 and w.sizes["dst_grid_size"] == 45
    np.testing.assert_allclose(w["dst_grid_frac"].values, 1.0, atol=1e-12)
    d, s = w["dst_address"].values - 1, w["src_address"].values - 1
    np.testing.assert_allclose(np.bincount(d, weights=w["remap_matrix"].values[:, 0], minlength=45), 1.0, atol=1e-12)
    got = np.bincount(s, weights=link_areas(w), minlength=60)
    assert np.abs(got - area_a).max() / area_a.mean() < 0.03
    # the identity: a mesh onto itself is the unit matrix (off-diagonal overlaps only from samples on an edge)
    same = gridgen.generate_weights(a, a, method="con")
    diag = same["dst_address"].values == same["src_address"].values
    assert same["remap_matrix"].values[diag, 0].min() > 0.999 and diag.sum() == 60


def fesom_grid():
    z = np.load(os.path.join(GOLDEN, "temp3d_fesom.npz"))
    g = Grid("points", z["lon"], z["lat"], name="fesom", cdo_type="unstructured")
    g.vertices = (z["lon_bnds"].astype(np.float64), z["lat_bnds"].astype(np.float64))
    return g, z


def test_the_reference_fesom_mesh_to_one_degree():
    """identity3d_test.py:28-32's pair (temp3d-fesom.nc -> r360x180, `con`): an ocean mesh, so land stays uncovered."""
    src, z = fesom_grid()
    w = gridgen.generate_weights(src, "r360x180", method="con")
    frac = w["dst_grid_frac"].values
    area = w["dst_grid_area"].values
    ocean_share = (frac * area).sum() / (4 * np.pi)
    assert 0.62 < ocean_share < 0.72                                   # the ocean is 71 % of the globe; coarse coasts
    assert ((frac == 0) | (frac > 0)).all() and frac.max() <= 1.0 and (frac == 0).sum() > 15000
    d, s = w["dst_address"].values - 1, w["src_address"].values - 1
    rows = np.bincount(d, weights=w["remap_matrix"].values[:, 0], minlength=frac.size)
    np.testing.assert_allclose(rows[frac > 0], 1.0, atol=1e-12)
    assert np.unique(s).size > 0.99 * src.size                         # (almost) every node feeds some target cell
    # a smooth field of the node positions comes back within half a cell (first order); surface temperature stays in its range
    y = np.bincount(d, weights=w["remap_matrix"].values[:, 0] * z["lat"][s], minlength=frac.size)
    dst = gridgen.parse_grid("r360x180")
    lat2d = np.repeat(dst.lat, 360)
    ok = frac > 0.999
    assert np.abs(y[ok] - lat2d[ok]).max() < 6.0 and np.abs(y[ok] - lat2d[ok]).mean() < 1.0   # cells of 0.4 - 9.5 degrees
    t = np.bincount(d, weights=w["remap_matrix"].values[:, 0] * z["temp"][0][s].astype(np.float64), minlength=frac.size)
    assert z["temp"][0].min() - 1e-9 <= t[frac > 0].min() and t.max() <= z["temp"][0].max() + 1e-9
    # the geometry is kept per target: a second mask costs no second search
    (kept,) = src._overlap_cache.values()
    gridgen.generate_weights(src, "r360x180", method="con", src_mask=(z["temp"][2] != 0))
    assert next(iter(src._overlap_cache.values())) is kept


def write_fesom_like_file(path, z, nt=2):
    """The fixture's mesh and levels in the layout of temp3d-fesom.nc: temp(time, nz1, nod2), lon / lat (nod2) naming
    their bounds (CF), bounds (nod2, 16) in degrees.  NetCDF-3 through scipy."""
    from scipy.io import netcdf_file
    n, V = z["lon_bnds"].shape
    with netcdf_file(path, "w") as nc:
        nc.createDimension("time", nt)
        nc.createDimension("nz1", z["nz1"].size)
        nc.createDimension("nod2", n)
        nc.createDimension("nv", V)
        for name, unit in (("lon", "degrees_east"), ("lat", "degrees_north")):
            v = nc.createVariable(name, "d", ("nod2",))
            v[:] = z[name]
            v.units, v.bounds = unit, name + "_bnds"
            b = nc.createVariable(name + "_bnds", "d", ("nod2", "nv"))
            b[:] = z[name + "_bnds"].astype(np.float64)
        v = nc.createVariable("nz1", "d", ("nz1",))
        v[:] = z["nz1"]
        v = nc.createVariable("time", "d", ("time",))
        v[:] = np.arange(nt, dtype=np.float64)
        v = nc.createVariable("temp", "f", ("time", "nz1", "nod2"))
        v[:] = np.stack([z["temp"] + np.float32(t) for t in range(nt)])
        v.coordinates = "lat lon"
        v.units = "degC"


def test_the_cell_polygons_of_a_file_reach_the_generator(tmp_path):
    from smmregrid_amd import CdoGenerate
    from smmregrid_amd.io import open_dataset
    _, z = fesom_grid()
    path = str(tmp_path / "fesom_like.nc")
    write_fesom_like_file(path, z)
    ds = open_dataset(path)
    g = CdoGenerate._grid_of(ds)
    assert g.kind == "points" and g.vertices is not None and g.vertices[0].shape == (3140, 16)
    np.testing.assert_allclose(g.vertices[1], z["lat_bnds"].astype(np.float64))
    # a field without its Dataset has centres only: `con` then has no cells to work with
    bare = CdoGenerate._grid_of(ds["temp"])
    assert bare.vertices is None
    with pytest.raises((ValueError, NotImplementedError)):
        gridgen.generate_weights(bare, "r36x18", method="con")


def test_cell_areas_of_polygon_grids():
    """areas_test.py:17-33 asks `CdoGenerate(file).areas()` for unstructured and curvilinear files: spherical polygon
    areas from the vertices (CDO's gridarea does the same for such grids)."""
    from smmregrid_amd import CdoGenerate, DataArray, Dataset
    src, areas = voronoi_mesh(200, seed=9, pad_to=15)
    got = gridgen.polygon_areas(*src.vertices)
    np.testing.assert_allclose(got, areas, rtol=1e-9)
    np.testing.assert_allclose(got.sum(), 4 * np.pi, rtol=1e-12)
    np.testing.assert_allclose(gridgen.polygon_areas(src.vertices[0][:, ::-1], src.vertices[1][:, ::-1]), areas, rtol=1e-9)
    # the reference's coarse FESOM mesh: an ocean (areas_test.py's OCEAN_SURFACE, for the finer tos-fesom.nc, is
    # 3.6e8 km2 +- 2 % of the Earth's surface; this mesh's coasts are coarser)
    fes, z = fesom_grid()
    lon = DataArray(z["lon"], dims=("nod2",), attrs={"units": "degrees_east", "bounds": "lon_bnds"})
    lat = DataArray(z["lat"], dims=("nod2",), attrs={"units": "degrees_north", "bounds": "lat_bnds"})
    ds = Dataset({"temp": DataArray(z["temp"][0], dims=("nod2",), coords={"lon": lon, "lat": lat}, name="temp"),
                  "lon_bnds": DataArray(z["lon_bnds"].astype(np.float64), dims=("nod2", "nv")),
                  "lat_bnds": DataArray(z["lat_bnds"].astype(np.float64), dims=("nod2", "nv"))})
    gen = CdoGenerate(ds, "r360x180", cdo="no-such-cdo-binary")
    a = gen.areas()
    assert a["cell_area"].shape == (3140,) and a["cell_area"].attrs["units"] == "m2"
    assert a["cell_area"].values.sum() / 1e6 == pytest.approx(3.5e8, abs=0.03 * 5.101e8)
    # ... and it is the area the conservative weights hand to the target grid (sampling error apart)
    w = gridgen.generate_weights(fes, "r360x180", method="con")
    covered = (w["dst_grid_frac"].values * w["dst_grid_area"].values).sum() * 6371000.0 ** 2
    assert covered == pytest.approx(a["cell_area"].values.sum(), rel=2e-3)
    # curvilinear cells keep their 2-D shape
    from tests.test_gridgen_curvilinear import rotated_pole_grid
    lon2, lat2, clon, clat = rotated_pole_grid(overlap=0)
    nav_lon = DataArray(lon2, dims=("y", "x"), attrs={"bounds": "bounds_nav_lon"})
    nav_lat = DataArray(lat2, dims=("y", "x"), attrs={"bounds": "bounds_nav_lat"})
    ds2 = Dataset({"tos": DataArray(np.zeros(lon2.shape), dims=("y", "x"), coords={"nav_lon": nav_lon, "nav_lat": nav_lat},
                                    name="tos"),
                   "bounds_nav_lon": DataArray(clon, dims=("y", "x", "nvertex")),
                   "bounds_nav_lat": DataArray(clat, dims=("y", "x", "nvertex"))})
    a2 = CdoGenerate(ds2, "r360x180", cdo="no-such-cdo-binary").areas()
    assert a2["cell_area"].shape == lon2.shape
    assert a2["cell_area"].values.sum() / 1e6 == pytest.approx(5.101e8, rel=5e-3)   # rotated parallels vs great circles
    bare = CdoGenerate(ds["temp"], "r360x180", cdo="no-such-cdo-binary")
    with pytest.raises(NotImplementedError):
        bare.areas()
