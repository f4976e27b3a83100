"""CPU tests pinning the oracle (oracle/oracle.py + oracle/oracle.c).

The reference stores no weights or golden outputs, so the oracle
is pinned by (i) two independent restatements agreeing, (ii) analytic known
answers that mirror the reference's own tests, (iii) the committed golden
fixtures in tests/golden/ (produced by tests/golden/make_golden.py).
"""
import os

import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import gridgen
from tests.helpers import assert_same, field, ragged_links, random_links

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def csr_of(ds):
    return oracle.coo_to_csr(ds.sizes["src_grid_size"], ds.sizes["dst_grid_size"],
                             ds["src_address"].values, ds["dst_address"].values,
                             ds["remap_matrix"].values)


# ----------------------------------------------------------------- operator build

def test_csr_c_matches_scipy_structure(rng):
    for n_src, n_dst, nnz in [(50, 40, 300), (1000, 333, 5000), (7, 3, 0), (1, 1, 5)]:
        src, dst, w = random_links(rng, n_src, n_dst, nnz, dup_frac=0.2)
        a = oracle.coo_to_csr(n_src, n_dst, src, dst, w)
        b = oracle.coo_to_csr_c(n_src, n_dst, src, dst, w)
        assert np.array_equal(a[0], b[0])
        assert np.array_equal(a[1], b[1])
        np.testing.assert_allclose(a[2], b[2], rtol=1e-14, atol=1e-300)


def test_duplicates_are_summed_and_zeros_kept():
    # two links on the same (dst, src) pair sum; an explicit zero link stays a link
    src = np.array([2, 2, 1, 3], np.int32)
    dst = np.array([1, 1, 1, 2], np.int32)
    w = np.array([0.25, 0.5, 0.0, 1.0])
    rowptr, col, val = oracle.coo_to_csr_c(3, 2, src, dst, w)
    assert rowptr.tolist() == [0, 2, 3]
    assert col.tolist() == [0, 1, 2]
    assert val.tolist() == [0.0, 0.75, 1.0]


def test_only_first_weight_column_is_used():
    # weights.py:33 -- remap_matrix[:, 0]; higher-order (bic/con2) columns are dropped
    rm = np.array([[0.5, 9.0, 9.0], [0.5, 9.0, 9.0]])
    csr = oracle.coo_to_csr(2, 1, [1, 2], [1, 1], rm)
    assert csr[2].tolist() == [0.5, 0.5]


def test_bad_address_rejected():
    with pytest.raises(ValueError):
        oracle.coo_to_csr_c(3, 2, np.array([4], np.int32), np.array([1], np.int32), np.array([1.0]))
    with pytest.raises(ValueError):
        oracle.coo_to_csr(3, 2, np.array([0]), np.array([1]), np.array([1.0]))


# ----------------------------------------------------------------- apply: two restatements agree

@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_apply_c_matches_numpy(rng, dtype):
    n_src, n_dst = 400, 150
    src, dst, w = random_links(rng, n_src, n_dst, 1200)
    csr = oracle.coo_to_csr(n_src, n_dst, src, dst, w)
    x = field(rng, 5, n_src, dtype=dtype, nan_frac=0.02, inf_frac=0.005)
    imask = (rng.random(n_dst) > 0.2).astype(np.int32)
    frac = rng.random(n_dst)
    for masked, amin in [(False, 0.0), (True, 0.0), (True, 0.5), (False, 0.9)]:
        a = oracle.apply(csr, x, masked, imask, frac, amin)
        b = oracle.apply_c(csr, x, masked, imask, frac, amin)
        assert_same(b, a, rtol=1e-12)


def test_apply_c_threads_identical(rng):
    src, dst, w = ragged_links(rng, 500, 300)
    csr = oracle.coo_to_csr_c(500, 300, src, dst, w)
    x = field(rng, 16, 500)
    a = oracle.apply_c(csr, x, threads=1)
    b = oracle.apply_c(csr, x, threads=4)
    assert_same(b, a, exact=True)


# ----------------------------------------------------------------- analytic known answers

def test_identity_weights_return_input(rng):
    n = 64
    csr = oracle.coo_to_csr_c(n, n, np.arange(1, n + 1, dtype=np.int32),
                              np.arange(1, n + 1, dtype=np.int32), np.ones(n))
    x = field(rng, 3, n)
    assert_same(oracle.apply_c(csr, x), x, exact=True)


def test_bilinear_reproduces_linear_field_in_latitude():
    w = gridgen.bilinear_weights("r180x90", "r90x45")     # BASELINE config 1 geometry
    csr = csr_of(w)
    src, dst = gridgen.parse_grid("r180x90"), gridgen.parse_grid("r90x45")
    slon, slat = src.centers()
    dlon, dlat = dst.centers()
    x = (3.0 * slat + 7.0)[None, :]
    y = oracle.apply_c(csr, x)
    np.testing.assert_allclose(y[0], 3.0 * dlat + 7.0, rtol=1e-12)
    # rows are stochastic: a constant field stays constant
    y1 = oracle.apply_c(csr, np.full((1, src.size), 273.15))
    np.testing.assert_allclose(y1, 273.15, rtol=1e-13)
    assert w.sizes["num_links"] == 4 * dst.size


def test_conservative_preserves_area_integral(rng):
    w = gridgen.conservative_weights("r72x36", "r24x12")
    csr = csr_of(w)
    src, dst = gridgen.parse_grid("r72x36"), gridgen.parse_grid("r24x12")
    x = field(rng, 2, src.size)
    y = oracle.apply_c(csr, x)

    def areas(g):
        return (np.diff(np.sin(np.radians(g.lat_b)))[:, None] * np.radians(np.diff(g.lon_b))[None, :]).ravel()
    np.testing.assert_allclose((y * areas(dst)).sum(axis=1), (x * areas(src)).sum(axis=1), rtol=1e-12)
    np.testing.assert_allclose(w["dst_grid_frac"].values, 1.0, rtol=1e-12)


def test_all_nan_timestep_stays_nan(rng):
    # mirrors tests/basic_test.py:31-39 of the reference
    w = gridgen.conservative_weights("r36x18", "r12x6")
    csr = csr_of(w)
    x = field(rng, 3, 36 * 18)
    x[1, :] = np.nan
    y = oracle.apply_c(csr, x)
    assert np.isnan(y[1]).all()
    assert np.isfinite(y[0]).all() and np.isfinite(y[2]).all()


def test_1e20_fill_quirk():
    # regrid.py:545-570: a missing source value becomes 1e20 before the product and only
    # results above 1e19 turn back into NaN: weight >= 0.1 -> NaN, weight < 0.1 -> finite ~ w*1e20
    csr = oracle.coo_to_csr_c(2, 2, np.array([1, 2, 1, 2], np.int32), np.array([1, 1, 2, 2], np.int32),
                              np.array([0.5, 0.5, 0.05, 0.95]))
    x = np.array([[np.nan, 1.0]])
    y = oracle.apply_c(csr, x)
    assert np.isnan(y[0, 0])
    assert y[0, 1] == 0.05 * 1e20 + 0.95 * 1.0
    # +inf is "invalid" too; float32 fields are filled with float32(1e20)
    y32 = oracle.apply_c(csr, np.array([[np.inf, 1.0]], dtype=np.float32))
    assert np.isnan(y32[0, 0])
    assert y32[0, 1] == 0.05 * float(np.float32(1e20)) + 0.95


@pytest.mark.parametrize("area_min", [0.0, 0.5, 0.75, 0.9])
def test_remap_area_min_thresholds(rng, area_min):
    # mirrors identity2d_test.py:40-45 / remapareamin_test.py:17-21: cells whose
    # unmasked fraction is below the threshold become NaN, monotonically more with the threshold
    src = gridgen.parse_grid("r72x36")
    mask = (rng.random(src.size) > 0.4).astype(np.int32)
    w = gridgen.conservative_weights(src, "r24x12", src_mask=mask)
    csr = csr_of(w)
    frac = w["dst_grid_frac"].values
    x = field(rng, 1, src.size)
    x[0, mask == 0] = np.nan
    imask = oracle.mask_apply_c(csr, mask)
    y = oracle.apply_c(csr, x, masked=True, dst_imask=imask, dst_frac=frac, area_min=area_min)
    expect_nan = (imask == 0) | ((frac < area_min) if area_min > 0 else False)
    assert np.array_equal(np.isnan(y[0]), expect_nan)


def test_mask_apply_threshold_and_check_mask(rng):
    csr = oracle.coo_to_csr_c(4, 3, np.array([1, 2, 3, 4, 1], np.int32),
                              np.array([1, 1, 2, 2, 3], np.int32), np.array([0.6, 0.4, 0.5, 0.5, 1.0]))
    m = oracle.mask_apply_c(csr, np.array([1, 0, 0, 1], np.int32))
    assert m.tolist() == [1, 1, 1]          # 0.6, 0.5 (not < 0.5), 1.0
    m = oracle.mask_apply_c(csr, np.array([0, 1, 0, 0], np.int32))
    assert m.tolist() == [0, 0, 0]
    assert np.array_equal(m, oracle.mask_apply(csr, np.array([0, 1, 0, 0])))
    assert oracle.check_mask(np.ones(5, int)) is False
    assert oracle.check_mask(np.array([1, 0, 1])) is True
    assert oracle.check_mask(np.array([[1, 1], [1, 0]])).tolist() == [False, True]


def test_level_selection_and_transpose(rng):
    # mirrors levels_test.py:10-27: data levels [14,15,17] / [15] pick the matching weights level
    wl = np.arange(10.0, 20.0)
    assert oracle.match_levels(wl, [14, 15, 17]).tolist() == [4, 5, 7]
    assert oracle.match_levels(wl, [15.0004]).tolist() == [5]
    with pytest.raises(ValueError):
        oracle.match_levels(wl, [15.5])
    S, D, L = 30, 12, 3
    csrs = []
    for _ in range(L):
        src, dst, w = random_links(rng, S, D, 60)
        csrs.append(oracle.coo_to_csr_c(S, D, src, dst, w))
    x = field(rng, 2 * L * 2, S).reshape(2, L, 2, S)
    imask = (rng.random((L, D)) > 0.3).astype(np.int32)
    out_t = oracle.apply_levels(csrs, x, 1, [2, 0, 1], [True, False, True], imask, None, 0.0, True)
    out_c = oracle.apply_levels(csrs, x, 1, [2, 0, 1], [True, False, True], imask, None, 0.0, False)
    assert out_t.shape == (2, 2, L, D) and out_c.shape == (L, 2, 2, D)
    assert_same(np.moveaxis(out_c, 0, -2), out_t, exact=True)
    # data level 1 uses weights level 0 (masked), data level 2 uses weights level 1 (not masked)
    ref = oracle.apply_c(csrs[0], x[:, 1].reshape(-1, S), masked=True, dst_imask=imask[0])
    assert_same(out_c[1].reshape(-1, D), ref, exact=True)
    ref = oracle.apply_c(csrs[1], x[:, 2].reshape(-1, S))
    assert_same(out_c[2].reshape(-1, D), ref, exact=True)


# ----------------------------------------------------------------- golden fixtures

@pytest.mark.parametrize("name", ["bil_r180x90_r90x45", "con_masked_levels", "ragged_random"])
def test_oracle_reproduces_golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    z = np.load(path)
    if "link_length" in z.files:
        L = z["link_length"].size
        csrs = [oracle.coo_to_csr_c(int(z["n_src"]), int(z["n_dst"]), z["src_address"][i, :z["link_length"][i]],
                                    z["dst_address"][i, :z["link_length"][i]],
                                    z["remap_matrix"][i, :z["link_length"][i]]) for i in range(L)]
        y = oracle.apply_levels(csrs, z["x"], 1, z["level_index"], z["masked_levels"],
                                z["dst_imask"], z["dst_frac"], float(z["area_min"]), True)
    else:
        csr = oracle.coo_to_csr_c(int(z["n_src"]), int(z["n_dst"]), z["src_address"], z["dst_address"],
                                  z["remap_matrix"])
        assert np.array_equal(csr[0], z["rowptr"]) and np.array_equal(csr[1], z["col"])
        y = oracle.apply_c(csr, z["x"], bool(z["masked"]), z["dst_imask"], z["dst_frac"],
                           float(z["area_min"]))
    assert_same(y, z["y"], exact=True)


# ----------------------------------------------------------------- the reference's own dask statements

@pytest.mark.parametrize("tag", ["float64", "float32"])
def test_oracle_matches_reference_dask_statements(tag):
    """tests/golden/dask_statements.npz holds outputs of the reference's statement sequence
    regrid.py:545-570 executed verbatim with dask.array 2021.10 (ma.fix_invalid / filled /
    tensordot / where) on a dense copy of the weights (make_dask_golden.py): NaN positions must
    match exactly, values to 1e-12 (dense BLAS vs sparse summation order)."""
    z = np.load(os.path.join(GOLDEN, "dask_statements.npz"))
    csr = oracle.coo_to_csr_c(int(z["n_src"]), int(z["n_dst"]), z["src_address"], z["dst_address"],
                              z["remap_matrix"])
    x = z["x_" + tag]
    for masked, amin in ((False, 0.0), (True, 0.0), (True, 0.5), (False, 0.9)):
        ref = z["y_%s_m%d_a%d" % (tag, int(masked), int(amin * 10))]
        for fn in (oracle.apply_c, oracle.apply):
            y = fn(csr, x.reshape(-1, x.shape[-1]), masked, z["dst_imask"], z["dst_frac"], amin)
            assert_same(y.reshape(ref.shape), ref, rtol=1e-12)
    for i in range(3):          # weights.py:47-52 executed with dask.array
        assert np.array_equal(oracle.mask_apply_c(csr, z["src_imask_%d" % i]), z["mask_tensordot_%d" % i])
        assert np.array_equal(oracle.mask_apply(csr, z["src_imask_%d" % i]), z["mask_tensordot_%d" % i])
    # the float32 field was filled with float32(1e20), not 1e20: a 0.05-weight link to a missing
    # value stays finite and carries exactly that constant
    assert np.isfinite(z["y_float32_m0_a0"]).any()


def test_oracle_output_independent_of_link_order(rng):
    """Same matrix, differently ordered / split link lists (what sparse.COO would canonicalise,
    weights.py:37-39): both restatements give the same CSR structure and outputs within 1e-12."""
    from smmregrid_amd import gridgen
    from tests.helpers import field, max_rel_spread, reorder_links
    w = gridgen.conservative_weights("r144x72", "r48x24")
    n_src, n_dst = 144 * 72, 48 * 24
    src, dst, ww = w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values[:, 0]
    x = field(rng, 5, n_src, nan_frac=0.01)
    outs_c, outs_py, structs = [], [], []
    for k in range(6):
        s2, d2, w2 = (src, dst, ww) if k == 0 else reorder_links(rng, src, dst, ww)
        csr_c = oracle.coo_to_csr_c(n_src, n_dst, s2, d2, w2)
        csr_py = oracle.coo_to_csr(n_src, n_dst, s2, d2, w2)
        assert np.array_equal(csr_c[0], csr_py[0]) and np.array_equal(csr_c[1], csr_py[1])
        structs.append((csr_c[0], csr_c[1]))
        outs_c.append(oracle.apply_c(csr_c, x, False, None, w["dst_grid_frac"].values, 0.5))
        outs_py.append(oracle.apply(csr_py, x, False, None, w["dst_grid_frac"].values, 0.5))
    for st in structs[1:]:
        assert np.array_equal(st[0], structs[0][0]) and np.array_equal(st[1], structs[0][1])
    assert max_rel_spread(outs_c + outs_py) <= 1e-12
