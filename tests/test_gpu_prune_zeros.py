"""SMM_CREATE_PRUNE_ZEROS: links of weight exactly zero are dropped when the operator is built.  The
reference multiplies them (sparse.COO keeps explicit zeros, weights.py:37-39); after the 1e20 fill
every gathered value is finite, so they add +-0.0 to a sum that starts at +0.0: outputs must be
bit-identical with and without them, NaN / inf inputs and negative fields included."""
import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import DataArray, Regridder, SparseOperator, _lib, gridgen, to_device
from tests.helpers import assert_same, field, random_links

pytestmark = pytest.mark.gpu


def test_aligned_bilinear_keeps_one_link_of_four(hip, rng):
    w = gridgen.bilinear_weights("r360x181", "r90x45")        # lon 1 deg -> 4 deg, lat nodes coincide
    args = (w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values, w["dst_address"].values,
            w["remap_matrix"].values)
    full = SparseOperator(*args, device=0)
    pruned = SparseOperator(*args, device=0, prune_zeros=True)
    n_zero = int((w["remap_matrix"].values[:, 0] == 0.0).sum())
    assert n_zero > 0 and pruned.nnz == full.nnz - n_zero and pruned.n_used_src < full.n_used_src
    rp, col, val = pruned.export_csr()
    assert (val != 0.0).all() and rp[-1] == pruned.nnz
    csr = full.export_csr()                                      # the oracle sees every link of the file
    for dtype in (np.float64, np.float32):
        x = field(rng, 40, full.n_src, dtype=dtype, nan_frac=0.03, inf_frac=0.01)
        x[3] = -np.abs(x[3])                                     # 0 * negative = -0.0 must not leak a sign
        x[4] = 0.0
        ref = oracle.apply_c(csr, x)
        for op in (full, pruned):
            for flags in (0, _lib.APPLY_KERNEL_SELL):
                assert_same(op.apply(to_device(x), flags=flags).to_host(), ref, exact=True)
        y = pruned.apply(to_device(x)).to_host()
        assert not np.signbit(y[np.isfinite(y) & (y == 0.0)]).any()
        assert_same(pruned.apply_host(x), ref, exact=True)
        assert_same(pruned.apply_sb(to_device(np.ascontiguousarray(x.T))).to_host(), ref, exact=True)
    m = (rng.random(full.n_src) > 0.4).astype(np.int32)
    assert np.array_equal(pruned.mask_apply(m), full.mask_apply(m))          # weights.py:47-52


def test_pruning_random_links_with_explicit_and_summed_zeros(hip, rng):
    n_src, n_dst = 3000, 400
    src, dst, w = random_links(rng, n_src, n_dst, 6000, zero_frac=0.3)
    # duplicates that cancel exactly: the SUMMED weight is what counts (COO sums duplicates first)
    src = np.concatenate([src, [7, 7]]).astype(np.int32)
    dst = np.concatenate([dst, [5, 5]]).astype(np.int32)
    w = np.concatenate([w, [0.25, -0.25]])
    full = SparseOperator(n_src, n_dst, src, dst, w, device=0)
    pruned = SparseOperator(n_src, n_dst, src, dst, w, device=0, prune_zeros=True, dst_dims=[20, 20])
    assert pruned.nnz < full.nnz and (pruned.export_csr()[2] != 0.0).all()
    x = field(rng, 33, n_src, nan_frac=0.05, inf_frac=0.01)
    assert_same(pruned.apply(to_device(x)).to_host(), oracle.apply_c(full.export_csr(), x), exact=True)
    with pytest.raises(_lib.SmmError):                          # unknown option bits are refused
        h = __import__("ctypes").c_void_p()
        _lib.call("smm_operator_create_opt", n_src, n_dst, 0, None, None, None, 1 << 12, 0,
                  __import__("ctypes").byref(h))


def test_regridder_option(hip, rng):
    g = gridgen.parse_grid("r360x181")
    x = 280.0 + 10.0 * rng.standard_normal((5, 181, 360))
    x[2, 50:60, 100:140] = np.nan
    fld = DataArray(x, dims=("time", "lat", "lon"), coords={"time": np.arange(5), "lat": g.lat, "lon": g.lon}, name="t")
    w = gridgen.bilinear_weights("r360x181", "r90x45")
    a = Regridder(weights=w, device=0).regrid(fld)
    rg = Regridder(weights=w, device=0, prune_zero_weights=True)
    b = rg.regrid(fld)
    assert rg.grids[0].weights_matrix.nnz < w.sizes["num_links"]
    assert_same(b.values, a.values, exact=True)
