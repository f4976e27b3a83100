"""Tile plans with blocks whose source footprint exceeds the LDS budget (polar caps of HEALPix
targets, folds of tripolar grids): those blocks gather straight from X inside the tile kernel."""
import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import OperatorGroup, SparseOperator, _lib, gridgen, to_device
from tests.helpers import assert_same, field, kernel_forms

pytestmark = pytest.mark.gpu


def wide_block_links(rng, n_src, n_dst, wide_rows, links_per_row, wide_links):
    """Compact 4-link stencils everywhere except `wide_rows`, whose links scatter over all of S."""
    src, dst, w = [], [], []
    for d in range(n_dst):
        if d in wide_rows:
            cols = rng.choice(n_src, size=wide_links, replace=False)
        else:
            c0 = (d * 7) % (n_src - 8)
            cols = c0 + np.arange(links_per_row)
        ww = rng.random(cols.size)
        src.append(cols + 1)
        dst.append(np.full(cols.size, d + 1))
        w.append(ww / ww.sum())
    return (np.concatenate(src).astype(np.int32), np.concatenate(dst).astype(np.int32), np.concatenate(w))


@pytest.mark.parametrize("wide_links,shape", [(12, "4-slice blocks"), (40, "single-wave blocks")])
def test_blocks_beyond_lds_budget_are_gathered_directly(hip, rng, wide_links, shape):
    n_src, n_dst = 200_000, 9000
    wide = set(range(256, 512)) | {1400}              # one whole 256-row block + a stray row
    src, dst, w = wide_block_links(rng, n_src, n_dst, wide, 4, wide_links)
    op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
    info = op.plan_info()
    assert info["tile_plan"], info                     # the plan survives: wide blocks are marked direct
    assert info["lds_bytes"] <= 65536
    csr = op.export_csr()
    imask = (rng.random(n_dst) > 0.1).astype(np.int32)
    frac = rng.random(n_dst)
    op.set_epilogue(imask, frac)
    for dtype in (np.float64, np.float32):
        x = field(rng, 9, n_src, dtype=dtype, nan_frac=0.01)
        ref = oracle.apply_c(csr, x, True, imask, frac, 0.3)
        y = op.apply(to_device(x), masked=True, remap_area_min=0.3, flags=_lib.APPLY_KERNEL_TILE).to_host()
        assert_same(y, ref, exact=True)
        ys = op.apply(to_device(x), masked=True, remap_area_min=0.3, flags=_lib.APPLY_KERNEL_SELL).to_host()
        assert_same(ys, ref, exact=True)


def test_too_many_direct_links_invalidate_the_plan(hip, rng):
    # every row scattered: more than a quarter of the links would be direct -> SELL serves better
    n_src, n_dst = 100_000, 512
    src, dst, w = wide_block_links(rng, n_src, n_dst, set(range(n_dst)), 4, 12)
    op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
    assert not op.plan_info()["tile_plan"]
    x = field(rng, 3, n_src)
    assert_same(op.apply(to_device(x)).to_host(), oracle.apply_c(op.export_csr(), x), exact=True)
    with pytest.raises(_lib.SmmError):
        op.apply(to_device(x), flags=_lib.APPLY_KERNEL_TILE)


def test_healpix_polar_blocks_in_group(hip, rng):
    # a HEALPix target from a coarse Gaussian source: polar pixels span all longitudes
    w = gridgen.bilinear_weights("F80", "hp64")
    S, D = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
    ops = [SparseOperator(S, D, w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values,
                          device=0) for _ in range(2)]
    grp = OperatorGroup(ops)
    x = field(rng, 2 * 2 * 3, S).reshape(3, 2, 2, S)
    y = grp.apply(to_device(x), np.array([1, 0], np.int32), transpose=True).to_host()
    ref = oracle.apply_levels([o.export_csr() for o in ops], x, 1, [1, 0], [False, False], None, None, 0.0, True)
    assert_same(y, ref, exact=True)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_tightened_budget_demotes_moderately_wide_blocks(hip, rng, dtype):
    """A block that would fit the full LDS budget but is far wider than all the others is demoted to
    direct gathering (tighten_tile_plan), so the LDS request -- and the workgroups per CU -- follow
    the typical block; the small tiles then run the multi-row steps.  Every kernel form stays exact."""
    n_dst, n_src = 256 * 150, 256 * 150 * 2 + 8
    d = np.repeat(np.arange(n_dst), 4)
    s_ = d * 2 + np.tile(np.arange(4), n_dst)
    wide = (d >= 256 * 50) & (d < 256 * 51)
    s_[wide] = 256 * 50 * 2 + rng.integers(0, 4800, size=int(wide.sum()))     # ~300 chunks, budget 512
    src, dst, w = (s_ + 1).astype(np.int32), (d + 1).astype(np.int32), rng.random(d.size)
    op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
    info = op.plan_info()
    assert info["tile_plan"] and info["lds_bytes"] <= 64 * 16 * 8, info        # 64 chunks, not ~300
    csr = op.export_csr()
    x = field(rng, 11, n_src, dtype=dtype, nan_frac=0.01)
    ref = oracle.apply_c(csr, x)
    for fl, knobs in kernel_forms({"tile_rows_per_step": 1}, {"xcd_run": -1}, {"xcd_run": 8}, {"tile_walk": 3}):
        with _lib.tuning(**knobs):
            assert_same(op.apply(to_device(x), flags=fl).to_host(), ref, exact=True)
