"""Seeded fuzz of the apply path against the CPU oracle: random sizes, link patterns, batch
shapes, dtypes, masks, thresholds, kernels and level groups -- all compared bit for bit."""
import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import OperatorGroup, SparseOperator, _lib, to_device
from tests.helpers import assert_same, field, kernel_forms, ragged_links, random_links

pytestmark = pytest.mark.gpu


def make_links(rng, kind, n_src, n_dst):
    if kind == "random":
        return random_links(rng, n_src, n_dst, int(rng.integers(0, 6 * n_dst + 1)), dup_frac=0.1)
    if kind == "ragged":
        return ragged_links(rng, n_src, n_dst, max_len=int(rng.integers(1, 70)))
    if kind == "longband":
        # long rows in windows that lie far apart (high-resolution source, coarse target): part-of-a-slice
        # blocks with rows split over lane groups, ragged lengths, some rows empty
        stride, max_len = int(rng.integers(10, 220)), int(rng.integers(20, 260))
        n_src_need = n_dst * stride + max_len + 1
        src, dst, w = [], [], []
        for d in range(n_dst):
            n = 0 if rng.random() < 0.1 else int(rng.integers(1, max_len + 1))
            cols = (d * stride + np.sort(rng.choice(max_len, size=n, replace=False))) % max(n_src, 1)
            cols = np.unique(cols)
            src.append(cols + 1)
            dst.append(np.full(cols.size, d + 1))
            w.append(rng.uniform(-0.2, 1.0, size=cols.size))
        perm = rng.permutation(sum(c.size for c in src))
        return (np.concatenate(src)[perm].astype(np.int32), np.concatenate(dst)[perm].astype(np.int32),
                np.concatenate(w)[perm])
    # banded stencil: k consecutive columns per row (structured grids), some rows empty
    k = int(rng.integers(1, 20))
    step = max(1, (n_src - k) // max(n_dst, 1))
    rows = np.flatnonzero(rng.random(n_dst) > 0.1)
    src = (rows[:, None] * step + np.arange(k)[None, :]) % n_src
    dst = np.repeat(rows, k)
    w = rng.uniform(-0.2, 1.0, size=src.size)
    perm = rng.permutation(src.size)
    return (src.ravel()[perm] + 1).astype(np.int32), (dst[perm] + 1).astype(np.int32), w[perm]


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_2d(hip, seed):
    rng = np.random.default_rng(1000 + seed)
    n_src = int(rng.integers(1, 6000))
    n_dst = int(rng.integers(1, 1500))
    kind = ["random", "ragged", "banded"][seed % 3]
    src, dst, w = make_links(rng, kind, n_src, n_dst)
    op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
    csr = op.export_csr()
    ref_csr = oracle.coo_to_csr(n_src, n_dst, src, dst, w)
    assert np.array_equal(csr[0], ref_csr[0]) and np.array_equal(csr[1], ref_csr[1])
    imask = (rng.random(n_dst) > 0.3).astype(np.int32)
    frac = rng.random(n_dst)
    op.set_epilogue(imask, frac)
    dtype = np.float32 if seed % 2 else np.float64
    n_batch = int(rng.integers(1, 40))
    x = field(rng, n_batch, n_src, dtype=dtype, nan_frac=0.03, inf_frac=0.005)
    masked = bool(seed % 2)
    amin = float(rng.choice([0.0, 0.25, 0.5, 0.9]))
    ref = oracle.apply_c(csr, x, masked, imask, frac, amin)
    kernels = [0, _lib.APPLY_KERNEL_SELL]
    if op.plan_info()["tile_plan"]:
        kernels.append(_lib.APPLY_KERNEL_TILE)
    for fl in kernels:
        y = op.apply(to_device(x), masked=masked, remap_area_min=amin, flags=fl).to_host()
        assert_same(y, ref, exact=True)
    yh = op.apply_host(x, masked=masked, remap_area_min=amin, chunk_rows=int(rng.integers(0, 9)))
    assert_same(yh, ref, exact=True)
    # the batch-fastest entry point (X transposed, full or packed to the used source cells)
    xt = np.ascontiguousarray(x.T)
    packed = bool(seed % 2)
    if packed:
        xt = np.ascontiguousarray(xt[op.used_sources()]) if op.n_used_src else np.zeros((0, n_batch), dtype)
    ysb = op.apply_sb(to_device(xt), masked=masked, remap_area_min=amin, packed=packed).to_host()
    assert_same(ysb, ref, exact=True)
    # mixed pinned / pageable buffers and the float32 narrowing store through the same pipeline
    from smmregrid_amd import pinned_empty
    xin = x
    if seed % 4 < 2:
        xin = pinned_empty(x.shape, x.dtype)
        xin[...] = x
    out_dtype = np.float32 if seed % 5 == 0 else np.float64
    yout = pinned_empty((n_batch, n_dst), out_dtype) if seed % 4 in (1, 2) else np.empty((n_batch, n_dst), out_dtype)
    op.apply_host(xin, out=yout, masked=masked, remap_area_min=amin, chunk_rows=int(rng.integers(0, 5)))
    assert_same(np.array(yout), ref.astype(out_dtype), exact=(out_dtype == np.float64), rtol=0)


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_long_rows(hip, seed):
    """Long rows in windows far apart: part-of-a-slice blocks, rows split over lane groups."""
    rng = np.random.default_rng(7000 + seed)
    n_src, n_dst = int(rng.integers(5000, 150000)), int(rng.integers(1, 700))
    src, dst, w = make_links(rng, "longband", n_src, n_dst)
    op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
    csr = op.export_csr()
    imask = (rng.random(n_dst) > 0.3).astype(np.int32)
    frac = rng.random(n_dst)
    op.set_epilogue(imask, frac)
    dtype = np.float32 if seed % 2 else np.float64
    x = field(rng, int(rng.integers(1, 30)), n_src, dtype=dtype, nan_frac=0.03, inf_frac=0.005)
    masked = bool(seed % 3)
    ref = oracle.apply_c(csr, x, masked, imask, frac, 0.5)
    forms = kernel_forms({"tile_split_rows": 1}, {"tile_walk": 3}, {"tile_links": 1}, {"tile_x_loads": 2})
    if not op.plan_info()["tile_plan"]:
        forms = [f for f in forms if f[0] != _lib.APPLY_KERNEL_TILE]
    for fl, knobs in forms:
        with _lib.tuning(**knobs):
            assert_same(op.apply(to_device(x), masked=masked, remap_area_min=0.5, flags=fl).to_host(), ref, exact=True)
    assert_same(op.apply_host(x, masked=masked, remap_area_min=0.5), ref, exact=True)


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_levels(hip, seed):
    rng = np.random.default_rng(5000 + seed)
    n_src = int(rng.integers(16, 3000)) * 2
    n_dst = int(rng.integers(1, 700))
    n_ops = int(rng.integers(1, 6))
    ops, csrs = [], []
    imask = (rng.random((n_ops, n_dst)) > 0.3).astype(np.int32)
    frac = rng.random((n_ops, n_dst))
    for i in range(n_ops):
        src, dst, w = make_links(rng, ["random", "ragged", "banded"][(seed + i) % 3], n_src, n_dst)
        op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
        op.set_epilogue(imask[i], frac[i])
        ops.append(op)
        csrs.append(op.export_csr())
    grp = OperatorGroup(ops)
    n_lev = int(rng.integers(1, 7))
    level_index = rng.integers(0, n_ops, size=n_lev).astype(np.int32)
    masked_levels = (rng.random(n_ops) > 0.4).astype(np.uint8)
    n_outer, n_inner = int(rng.integers(1, 6)), int(rng.integers(1, 4))
    x = field(rng, n_outer * n_lev * n_inner, n_src, nan_frac=0.02).reshape(n_outer, n_lev, n_inner, n_src)
    amin = float(rng.choice([0.0, 0.5]))
    for transpose in (True, False):
        ref = oracle.apply_levels(csrs, x, 1, level_index, masked_levels.astype(bool), imask, frac, amin, transpose)
        flags = [0, _lib.APPLY_KERNEL_SELL] + ([_lib.APPLY_KERNEL_TILE] if grp.plan_info()["tile_plan"] else [])
        for fl in flags:
            y = grp.apply(to_device(x), level_index, masked_levels, masked=True, remap_area_min=amin,
                          transpose=transpose, flags=fl).to_host()
            assert_same(y, ref, exact=True)
        yh = grp.apply_host(x, level_index, masked_levels, masked=True, remap_area_min=amin, transpose=transpose,
                            chunk_outer=int(rng.integers(0, 4)))
        assert_same(yh, ref, exact=True)


@pytest.mark.parametrize("n_src", [1, 2, 3, 5])
def test_source_rows_shorter_than_a_staging_piece(hip, n_src):
    """S below one 16-B piece (4 f32 / 2 f64 elements): the tile kernel cannot stage such rows and
    gathers them directly."""
    rng = np.random.default_rng(n_src)
    n_dst = 70
    src = rng.integers(1, n_src + 1, size=150).astype(np.int32)
    dst = rng.integers(1, n_dst + 1, size=150).astype(np.int32)
    w = rng.random(150)
    op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
    for dtype in (np.float32, np.float64):
        x = field(rng, 6, n_src, dtype=dtype)
        ref = oracle.apply_c(op.export_csr(), x)
        flags = [0, _lib.APPLY_KERNEL_SELL] + ([_lib.APPLY_KERNEL_TILE] if op.plan_info()["tile_plan"] else [])
        for fl in flags:
            assert_same(op.apply(to_device(x), flags=fl).to_host(), ref, exact=True)
