"""Rows with very wide source footprints (high-resolution source, coarse target: ~100 links per
row, 10 source rows under every destination cell): a whole 64-row slice no longer fits the LDS
budget, so the tile plan gives a block 32 / 16 / 8 rows of the slice."""
import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import OperatorGroup, SparseOperator, _lib, gridgen, to_device
from tests.helpers import assert_same, field, kernel_forms

pytestmark = pytest.mark.gpu


def build(sgrid, tgrid, mask=None):
    w = gridgen.generate_weights(sgrid, tgrid, method="con", src_mask=mask)
    op = SparseOperator(w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values,
                        w["dst_address"].values, w["remap_matrix"].values, device=0)
    op.set_epilogue(w["dst_grid_imask"].values, w["dst_grid_frac"].values)
    return w, op


@pytest.mark.parametrize("sgrid,tgrid", [("r720x360", "r72x36"), ("r1800x900", "r180x90")])
def test_part_slice_blocks_match_the_oracle(hip, rng, sgrid, tgrid):
    w, op = build(sgrid, tgrid)
    info = op.plan_info()
    assert op.max_row_nnz >= 100
    assert info["tile_plan"] and info["tile_preferred"] and info["rows_per_block"] in (32, 16, 8), info
    csr = op.export_csr()
    imask, frac = w["dst_grid_imask"].values, w["dst_grid_frac"].values
    for dtype in (np.float64, np.float32):
        x = field(rng, 13, op.n_src, dtype=dtype, nan_frac=0.01)
        ref = oracle.apply_c(csr, x, True, imask, frac, 0.5)
        for fl, knobs in kernel_forms({"xcd_run": -1}, {"tile_walk": 5}, {"tile_split_rows": 1}, {"tile_links": 1}):
            with _lib.tuning(**knobs):
                assert_same(op.apply(to_device(x), masked=True, remap_area_min=0.5, flags=fl).to_host(), ref, exact=True)
    assert_same(op.apply_host(x, masked=True, remap_area_min=0.5), ref, exact=True)


def test_group_of_part_slice_and_full_slice_operators(hip, rng):
    """Levels with different native block shapes share the most finely split one."""
    nx, ny = 720, 360
    mask = (rng.random(nx * ny) > 0.4).astype(np.int32)
    w0, op0 = build("r720x360", "r72x36")
    w1, op1 = build("r720x360", "r72x36", mask=mask)
    grp = OperatorGroup([op0, op1])
    assert grp.plan_info()["tile_plan"]
    x = field(rng, 3 * 2 * 2, op0.n_src, nan_frac=0.02).reshape(3, 2, 2, op0.n_src)
    level_index = np.array([1, 0], dtype=np.int32)
    masked_levels = np.array([1, 1], dtype=np.uint8)
    imask = np.stack([w0["dst_grid_imask"].values, w1["dst_grid_imask"].values])
    frac = np.stack([w0["dst_grid_frac"].values, w1["dst_grid_frac"].values])
    ref = oracle.apply_levels([op0.export_csr(), op1.export_csr()], x, 1, level_index, masked_levels.astype(bool),
                              imask, frac, 0.5, True)
    for fl in (0, _lib.APPLY_KERNEL_TILE, _lib.APPLY_KERNEL_SELL):
        y = grp.apply(to_device(x), level_index, masked_levels, masked=True, remap_area_min=0.5, flags=fl).to_host()
        assert_same(y, ref, exact=True)


@pytest.mark.parametrize("stride,max_len,max_rows", [(200, 190, 8), (120, 150, 16), (50, 90, 32)])
def test_split_rows_of_ragged_length(hip, rng, stride, max_len, max_rows):
    """Rows of very different length (0 .. max_len links, some empty) inside part-of-a-slice blocks:
    the lane groups' shares follow the block's longest row, shorter rows leave groups empty.
    (Which part of a slice a block gets follows the footprint of the random rows: the widest case
    must end up with 8-row blocks, the others with at most 16 / 32 rows.)"""
    n_dst = 333
    n_src = n_dst * stride + max_len + 7
    src, dst, w = [], [], []
    for d in range(n_dst):
        n = 0 if d % 11 == 3 else int(rng.integers(1, max_len + 1))
        cols = d * stride + np.sort(rng.choice(max_len, size=n, replace=False))
        src.append(cols + 1)
        dst.append(np.full(n, d + 1))
        w.append(rng.uniform(-0.3, 1.0, size=n))
    src, dst, w = (np.concatenate(src).astype(np.int32), np.concatenate(dst).astype(np.int32), np.concatenate(w))
    op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
    info = op.plan_info()
    assert info["tile_plan"] and 8 <= info["rows_per_block"] <= max_rows, info
    imask = (rng.random(n_dst) > 0.2).astype(np.int32)
    frac = rng.random(n_dst)
    op.set_epilogue(imask, frac)
    csr = op.export_csr()
    for dtype in (np.float64, np.float32):
        x = field(rng, 9, n_src, dtype=dtype, nan_frac=0.02, inf_frac=0.002)
        ref = oracle.apply_c(csr, x, True, imask, frac, 0.4)
        for fl, knobs in kernel_forms({"tile_split_rows": 1}, {"tile_walk": 3}, {"tile_links": 1}):
            with _lib.tuning(**knobs):
                assert_same(op.apply(to_device(x), masked=True, remap_area_min=0.4, flags=fl).to_host(), ref, exact=True)


def test_split_kernel_with_direct_blocks(hip, rng):
    """One 16-row block whose links scatter over the whole source (beyond the LDS budget even for
    part-of-a-slice blocks) is gathered directly inside the split-row kernel."""
    stride, max_len, n_dst = 120, 150, 400
    n_src = n_dst * stride + max_len + 7
    src, dst, w = [], [], []
    for d in range(n_dst):
        n = int(rng.integers(1, max_len + 1))
        if 160 <= d < 176:                                     # the scattered block
            cols = np.sort(rng.choice(n_src, size=n, replace=False))
        else:
            cols = d * stride + np.sort(rng.choice(max_len, size=n, replace=False))
        src.append(cols + 1)
        dst.append(np.full(n, d + 1))
        w.append(rng.uniform(-0.3, 1.0, size=n))
    src, dst, w = (np.concatenate(src).astype(np.int32), np.concatenate(dst).astype(np.int32), np.concatenate(w))
    op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
    info = op.plan_info()
    assert info["tile_plan"] and info["rows_per_block"] == 16, info
    imask = (rng.random(n_dst) > 0.2).astype(np.int32)
    frac = rng.random(n_dst)
    op.set_epilogue(imask, frac)
    csr = op.export_csr()
    for dtype in (np.float64, np.float32):
        x = field(rng, 7, n_src, dtype=dtype, nan_frac=0.02)
        ref = oracle.apply_c(csr, x, True, imask, frac, 0.4)
        for fl, knobs in kernel_forms({"tile_split_rows": 1}, sell_knobs=()):
            with _lib.tuning(**knobs):
                assert_same(op.apply(to_device(x), masked=True, remap_area_min=0.4, flags=fl).to_host(), ref, exact=True)
