"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on
the same seeded inputs.  Bar (north star): CSR indices bit-exact; f64 values
within 1e-6 relative with identical NaN positions -- the kernels keep the
oracle's summation order without FMA contraction, so f64 outputs are in fact
compared bit for bit (exact=True) wherever both sides compute in f64.
"""
import os

import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import OperatorGroup, SparseOperator, _lib, gridgen, to_device
from tests.helpers import assert_same, field, ragged_links, random_links

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-6  # north-star tolerance for f64 values


def make_op(n_src, n_dst, src, dst, w):
    return SparseOperator(n_src, n_dst, src, dst, w, device=0)


def run(op, x, masked=False, amin=0.0, flags=0, out_dtype=np.float64):
    dx = to_device(x)
    dy = op.apply(dx, masked=masked, remap_area_min=amin, flags=flags, out_dtype=out_dtype)
    return dy.to_host()


KERNELS = [("sell", _lib.APPLY_KERNEL_SELL), ("tile", _lib.APPLY_KERNEL_TILE)]


def need_kernel(op, kname):
    """Skip a tile-kernel case when the operator has no LDS plan (a block's source
    footprint exceeds the LDS budget): the library refuses the forced flag there."""
    ops = op if isinstance(op, (list, tuple)) else [op]
    if kname == "tile" and any(not o.plan_info()["tile_plan"] for o in ops):
        pytest.skip("tile kernel not applicable to this operator")


# ----------------------------------------------------------------- operator build (K6)

def test_csr_export_bit_exact(hip, rng):
    for n_src, n_dst, nnz in [(50, 40, 300), (1000, 333, 5000), (7, 3, 0), (1, 1, 5), (4097, 129, 20000)]:
        src, dst, w = random_links(rng, n_src, n_dst, nnz, dup_frac=0.2)
        op = make_op(n_src, n_dst, src, dst, w)
        rowptr, col, val = op.export_csr()
        a = oracle.coo_to_csr(n_src, n_dst, src, dst, w)      # scipy
        b = oracle.coo_to_csr_c(n_src, n_dst, src, dst, w)    # sequential C
        assert np.array_equal(rowptr, a[0]) and np.array_equal(col, a[1])
        assert np.array_equal(val.view(np.uint64), b[2].view(np.uint64))
        np.testing.assert_allclose(val, a[2], rtol=1e-14, atol=1e-300)
        assert op.nnz == a[1].size
        assert op.n_used_src == np.unique(a[1]).size


def test_invalid_links_rejected(hip):
    with pytest.raises(_lib.SmmError) as e:
        make_op(3, 2, np.array([4], np.int32), np.array([1], np.int32), np.array([1.0]))
    assert e.value.code == _lib.SMM_ERR_INVALID
    with pytest.raises(_lib.SmmError):
        make_op(3, 2, np.array([1], np.int32), np.array([0], np.int32), np.array([1.0]))


# ----------------------------------------------------------------- 2-D apply (K1+K2+K3)

@pytest.mark.parametrize("kname,kflag", KERNELS)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_apply_random_matrix(hip, rng, dtype, kname, kflag):
    n_src, n_dst = 4096, 1000          # D not a multiple of 64 / 256
    src, dst, w = random_links(rng, n_src, n_dst, 6000)
    op = make_op(n_src, n_dst, src, dst, w)
    csr = op.export_csr()
    imask = (rng.random(n_dst) > 0.2).astype(np.int32)
    frac = rng.random(n_dst)
    op.set_epilogue(imask, frac)
    need_kernel(op, kname)
    for n_batch in (1, 2, 3, 5, 8, 9, 17, 33):
        x = field(rng, n_batch, n_src, dtype=dtype, nan_frac=0.02, inf_frac=0.004)
        for masked, amin in [(False, 0.0), (True, 0.0), (True, 0.5), (False, 0.9)]:
            y = run(op, x, masked, amin, kflag)
            ref = oracle.apply_c(csr, x, masked, imask, frac, amin)
            assert_same(y, ref, exact=True)
            assert_same(y, oracle.apply(csr, x, masked, imask, frac, amin), rtol=RTOL)


@pytest.mark.parametrize("kname,kflag", KERNELS)
def test_apply_structured_weights(hip, rng, kname, kflag):
    cases = [("bil", "r180x90", "r90x45"), ("con", "r144x72", "r36x18"), ("bil", "r96x48", "hp8"),
             ("nn", "r64x32", "r20x10"), ("con", "r90x45", "r120x60")]
    for method, s, d in cases:
        w = gridgen.generate_weights(s, d, method=method)
        n_src, n_dst = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
        op = make_op(n_src, n_dst, w["src_address"].values, w["dst_address"].values,
                     w["remap_matrix"].values)
        op.set_epilogue(w["dst_grid_imask"].values, w["dst_grid_frac"].values)
        csr = op.export_csr()
        need_kernel(op, kname)
        x = field(rng, 11, n_src, nan_frac=0.01)
        y = run(op, x, False, 0.5, kflag)
        assert_same(y, oracle.apply_c(csr, x, False, None, w["dst_grid_frac"].values, 0.5), exact=True)


@pytest.mark.parametrize("kname,kflag", KERNELS)
def test_apply_ragged_rows(hip, rng, kname, kflag):
    n_src, n_dst = 3000, 777
    src, dst, w = ragged_links(rng, n_src, n_dst, max_len=30)
    op = make_op(n_src, n_dst, src, dst, w)
    need_kernel(op, kname)
    csr = op.export_csr()
    x = field(rng, 7, n_src, nan_frac=0.05)
    y = run(op, x, flags=kflag)
    assert_same(y, oracle.apply_c(csr, x), exact=True)


@pytest.mark.parametrize("kname,kflag", KERNELS)
def test_long_rows(hip, rng, kname, kflag):
    # rows longer than 32 links: the tile kernel streams the links from L2 instead of registers
    n_src, n_dst = 2000, 300
    src, dst, w = ragged_links(rng, n_src, n_dst, max_len=120)
    op = make_op(n_src, n_dst, src, dst, w)
    assert op.max_row_nnz > 32
    need_kernel(op, kname)
    x = field(rng, 5, n_src, nan_frac=0.02)
    assert_same(run(op, x, flags=kflag), oracle.apply_c(op.export_csr(), x), exact=True)


@pytest.mark.parametrize("kname,kflag", KERNELS)
def test_odd_source_size_rows_not_16B_aligned(hip, rng, kname, kflag):
    # S odd: f64 batch rows start on 8-byte, f32 rows on 4-byte boundaries; the 16-B staging loads
    # of the tile kernel only need element alignment
    n_src, n_dst = 1001, 300
    src, dst, w = random_links(rng, n_src, n_dst, 2000)
    op = make_op(n_src, n_dst, src, dst, w)
    need_kernel(op, kname)
    for dtype in (np.float64, np.float32):
        x = field(rng, 5, n_src, dtype=dtype, nan_frac=0.02)
        assert_same(run(op, x, flags=kflag), oracle.apply_c(op.export_csr(), x), exact=True)


@pytest.mark.parametrize("kname,kflag", KERNELS)
def test_tail_chunk_at_end_of_row(hip, rng, kname, kflag):
    # links on the last source cells: the staged chunk is clipped at n_src (S % 16 != 0)
    n_src, n_dst = 1000 + 6, 64
    src = np.concatenate([np.full(n_dst, n_src), np.full(n_dst, n_src - 1), np.arange(1, n_dst + 1)]).astype(np.int32)
    dst = np.tile(np.arange(1, n_dst + 1), 3).astype(np.int32)
    w = rng.random(src.size)
    op = make_op(n_src, n_dst, src, dst, w)
    x = field(rng, 3, n_src)
    assert_same(run(op, x, flags=kflag), oracle.apply_c(op.export_csr(), x), exact=True)


def test_empty_operator_and_empty_batch(hip, rng):
    op = make_op(10, 5, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0))
    x = field(rng, 2, 10)
    assert np.array_equal(run(op, x), np.zeros((2, 5)))
    y = run(op, np.zeros((0, 10)))
    assert y.shape == (0, 5)


def test_1e20_quirk_and_f32_fill(hip):
    op = make_op(2, 2, np.array([1, 2, 1, 2], np.int32), np.array([1, 1, 2, 2], np.int32),
                 np.array([0.5, 0.5, 0.05, 0.95]))
    y = run(op, np.array([[np.nan, 1.0]]))
    assert np.isnan(y[0, 0]) and y[0, 1] == 0.05 * 1e20 + 0.95
    y32 = run(op, np.array([[-np.inf, 1.0]], dtype=np.float32))
    assert np.isnan(y32[0, 0]) and y32[0, 1] == 0.05 * float(np.float32(1e20)) + 0.95
    # NO_FILL leaves NaN to propagate through the product
    yn = run(op, np.array([[np.nan, 1.0]]), flags=_lib.APPLY_NO_FILL)
    assert np.isnan(yn).all()


def test_f32_output_is_rounded_f64_result(hip, rng):
    n_src, n_dst = 512, 200
    src, dst, w = random_links(rng, n_src, n_dst, 900)
    op = make_op(n_src, n_dst, src, dst, w)
    x = field(rng, 4, n_src, dtype=np.float32, nan_frac=0.01)
    y = run(op, x, out_dtype=np.float32)
    ref = oracle.apply_c(op.export_csr(), x).astype(np.float32)
    assert y.dtype == np.float32
    assert_same(y, ref, rtol=0)


def test_argument_errors(hip, rng):
    op = make_op(8, 4, np.array([1], np.int32), np.array([1], np.int32), np.array([1.0]))
    dx = to_device(np.zeros((2, 8)))
    with pytest.raises(_lib.SmmError):       # masked without dst_imask
        op.apply(dx, masked=True)
    with pytest.raises(_lib.SmmError):       # area_min without dst_frac
        op.apply(dx, remap_area_min=0.5)
    op.set_epilogue(np.ones(4, np.int32), np.ones(4))
    with pytest.raises(_lib.SmmError):       # regrid.py:124-125
        op.apply(dx, remap_area_min=1.5)
    with pytest.raises(ValueError):
        op.apply(to_device(np.zeros((2, 7))))


# ----------------------------------------------------------------- mask pre-compute (K5)

def test_mask_apply_matches_oracle(hip, rng):
    src_g = gridgen.parse_grid("r72x36")
    mask = (rng.random(src_g.size) > 0.45).astype(np.int32)
    w = gridgen.conservative_weights(src_g, "r24x12", src_mask=mask)
    op = make_op(src_g.size, 24 * 12, w["src_address"].values, w["dst_address"].values,
                 w["remap_matrix"].values)
    got = op.mask_apply(mask)
    assert np.array_equal(got, oracle.mask_apply_c(op.export_csr(), mask))
    assert np.array_equal(got, oracle.mask_apply(op.export_csr(), mask))
    # threshold edge: exactly 0.5 is NOT below 0.5
    op2 = make_op(2, 1, np.array([1, 2], np.int32), np.array([1, 1], np.int32), np.array([0.5, 0.5]))
    assert op2.mask_apply(np.array([1, 0], np.int32)).tolist() == [1]


# ----------------------------------------------------------------- grouped levels (K4)

@pytest.mark.parametrize("kname,kflag", KERNELS)
@pytest.mark.parametrize("transpose", [True, False])
def test_group_apply_random(hip, rng, transpose, kname, kflag):
    S, D, L = 640, 200, 5
    ops, csrs = [], []
    imask = (rng.random((L, D)) > 0.3).astype(np.int32)
    frac = rng.random((L, D))
    for l in range(L):
        src, dst, w = random_links(rng, S, D, 300 + 150 * l)
        op = make_op(S, D, src, dst, w)
        op.set_epilogue(imask[l], frac[l])
        ops.append(op)
        csrs.append(op.export_csr())
    grp = OperatorGroup(ops)
    need_kernel(grp, kname)
    level_index = np.array([3, 0, 4, 4], np.int32)          # 4 data levels, sub-selection + repeat
    masked_levels = np.array([1, 0, 1, 1, 0], np.uint8)
    for n_outer, n_inner in [(3, 1), (2, 3), (1, 1), (9, 2)]:
        x = field(rng, n_outer * 4 * n_inner, S, nan_frac=0.03).reshape(n_outer, 4, n_inner, S)
        dy = grp.apply(to_device(x), level_index, masked_levels, masked=True, remap_area_min=0.4,
                       transpose=transpose, flags=kflag)
        ref = oracle.apply_levels(csrs, x, 1, level_index, masked_levels.astype(bool), imask, frac,
                                  0.4, transpose)
        assert_same(dy.to_host(), ref, exact=True)
    with pytest.raises(_lib.SmmError):
        grp.apply(to_device(x), np.array([0, 1, 2, 7], np.int32), masked_levels)


# ----------------------------------------------------------------- golden fixtures

@pytest.mark.parametrize("kname,kflag", KERNELS)
@pytest.mark.parametrize("name", ["bil_r180x90_r90x45", "ragged_random"])
def test_golden_2d(hip, name, kname, kflag):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    op = make_op(int(z["n_src"]), int(z["n_dst"]), z["src_address"], z["dst_address"], z["remap_matrix"])
    rowptr, col, val = op.export_csr()
    assert np.array_equal(rowptr, z["rowptr"]) and np.array_equal(col, z["col"])
    assert np.array_equal(val.view(np.uint64), z["val"].view(np.uint64))
    op.set_epilogue(z["dst_imask"], z["dst_frac"])
    need_kernel(op, kname)
    y = run(op, z["x"], bool(z["masked"]), float(z["area_min"]), kflag)
    assert_same(y, z["y"], exact=True)


@pytest.mark.parametrize("kname,kflag", KERNELS)
def test_golden_masked_levels(hip, kname, kflag):
    z = np.load(os.path.join(GOLDEN, "con_masked_levels.npz"))
    ll = z["link_length"]
    ops = []
    for i in range(ll.size):
        op = make_op(int(z["n_src"]), int(z["n_dst"]), z["src_address"][i, :ll[i]],
                     z["dst_address"][i, :ll[i]], z["remap_matrix"][i, :ll[i]])
        assert np.array_equal(op.mask_apply(z["src_imask"][i]), z["dst_imask"][i])
        op.set_epilogue(z["dst_imask"][i], z["dst_frac"][i])
        ops.append(op)
    grp = OperatorGroup(ops)
    need_kernel(grp, kname)
    x = z["x"]                                               # (T, L, S)
    T, L, S = x.shape
    dy = grp.apply(to_device(x.reshape(T, L, 1, S)), z["level_index"],
                   z["masked_levels"].astype(np.uint8), masked=True,
                   remap_area_min=float(z["area_min"]), transpose=True, flags=kflag)
    assert_same(dy.to_host().reshape(z["y"].shape), z["y"], exact=True)


# ----------------------------------------------------------------- full-size properties (config 2 geometry)

def test_config2_geometry_properties(hip, rng):
    """r1440x721 -> r360x180 bilinear at full grid size, reduced batch: size-independent
    properties -- row-stochastic weights keep a constant field constant, the operator is
    linear, and both kernels agree bit for bit."""
    w = gridgen.bilinear_weights("r1440x721", "r360x180")
    S, D = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
    assert (S, D, w.sizes["num_links"]) == (1038240, 64800, 259200)
    op = make_op(S, D, w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values)
    assert op.n_used_src == 259200
    B = 24
    x1 = field(rng, B, S)
    x2 = field(rng, B, S)
    d1, d2, d3 = to_device(x1), to_device(x2), to_device(2.0 * x1 + x2)
    y1 = op.apply(d1, flags=_lib.APPLY_KERNEL_SELL).to_host()
    y1t = op.apply(d1, flags=_lib.APPLY_KERNEL_TILE).to_host()
    assert_same(y1t, y1, exact=True)
    y2 = op.apply(d2).to_host()
    y3 = op.apply(d3).to_host()
    np.testing.assert_allclose(y3, 2.0 * y1 + y2, rtol=1e-12)
    yc = op.apply(to_device(np.full((2, S), 287.5))).to_host()
    np.testing.assert_allclose(yc, 287.5, rtol=1e-14)
    # spot parity against the oracle on a row subset
    ref = oracle.apply_c(op.export_csr(), x1[:3])
    assert_same(y1[:3], ref, exact=True)


# ----------------------------------------------------------------- host-buffer streaming pipeline

@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_apply_host_pipeline_matches_device_path(hip, rng, dtype):
    from smmregrid_amd import pinned_empty
    n_src, n_dst = 3001, 777                      # odd S: device rows are re-pitched to 16 B
    src, dst, w = random_links(rng, n_src, n_dst, 5000)
    op = make_op(n_src, n_dst, src, dst, w)
    imask = (rng.random(n_dst) > 0.2).astype(np.int32)
    frac = rng.random(n_dst)
    op.set_epilogue(imask, frac)
    csr = op.export_csr()
    x = field(rng, 37, n_src, dtype=dtype, nan_frac=0.02)
    ref = oracle.apply_c(csr, x, True, imask, frac, 0.4)
    for chunk in (0, 1, 5, 16, 64):               # single chunk, many chunks, ragged tail
        y = op.apply_host(x, masked=True, remap_area_min=0.4, chunk_rows=chunk)
        assert_same(y, ref, exact=True)
    # pinned input and output buffers are DMA'd directly
    xp = pinned_empty(x.shape, dtype)
    xp[...] = x
    yp = pinned_empty((37, n_dst), np.float64)
    op.apply_host(xp, out=yp, masked=True, remap_area_min=0.4, chunk_rows=7)
    assert_same(np.array(yp), ref, exact=True)
    # a row-strided view (ldx > S) is consumed without a copy
    wide = np.zeros((37, n_src + 13), dtype=dtype)
    wide[:, :n_src] = x
    y = op.apply_host(wide[:, :n_src], masked=True, remap_area_min=0.4, chunk_rows=9)
    assert_same(y, ref, exact=True)
    assert op.apply_host(np.zeros((0, n_src), dtype)).shape == (0, n_dst)


def test_apply_is_reentrant_across_threads_and_streams(hip, rng):
    """Handles are immutable after set_epilogue: concurrent smm_apply calls from several host
    threads on their own streams (the reference's dask threaded scheduler does this,
    regrid.py:29-30) give the same bits as a serial call."""
    import threading
    from smmregrid_amd.device import Stream
    w = gridgen.conservative_weights("r144x72", "r48x24")
    op = make_op(144 * 72, 48 * 24, w["src_address"].values, w["dst_address"].values,
                 w["remap_matrix"].values)
    op.set_epilogue(w["dst_grid_imask"].values, w["dst_grid_frac"].values)
    xs = [field(rng, 16, 144 * 72, nan_frac=0.01) for _ in range(6)]
    refs = [oracle.apply_c(op.export_csr(), x, False, None, w["dst_grid_frac"].values, 0.5) for x in xs]
    outs = [None] * len(xs)
    errors = []

    def work(i):
        try:
            s = Stream()
            dx = to_device(xs[i])
            for _ in range(20):
                dy = op.apply(dx, remap_area_min=0.5, stream=s)
            s.synchronize()
            outs[i] = dy.to_host()
        except Exception as exc:  # pragma: no cover
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(xs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    for o, r in zip(outs, refs):
        assert_same(o, r, exact=True)


@pytest.mark.parametrize("kname,kflag", KERNELS)
def test_row_pitch_larger_than_grid(hip, rng, kname, kflag):
    """smm_apply with ldx > S and ldy > D (fields embedded in wider device buffers)."""
    import ctypes
    from smmregrid_amd.device import DeviceArray
    n_src, n_dst, B = 1024, 300, 6
    src, dst, w = random_links(rng, n_src, n_dst, 1500)
    op = make_op(n_src, n_dst, src, dst, w)
    need_kernel(op, kname)
    ldx, ldy = n_src + 16, n_dst + 5
    xw = np.full((B, ldx), np.nan)
    xw[:, :n_src] = field(rng, B, n_src)
    yw = np.full((B, ldy), -7.0)
    dx, dy = to_device(xw), to_device(yw)
    _lib.call("smm_apply", op.handle, ctypes.c_void_p(dx.ptr), _lib.SMM_F64, ldx, ctypes.c_void_p(dy.ptr),
              _lib.SMM_F64, ldy, B, 0.0, kflag, None)
    got = dy.to_host()
    assert_same(got[:, :n_dst], oracle.apply_c(op.export_csr(), np.ascontiguousarray(xw[:, :n_src])), exact=True)
    assert (got[:, n_dst:] == -7.0).all()          # the padding columns of Y are untouched
    with pytest.raises(_lib.SmmError):             # pitch smaller than the grid
        _lib.call("smm_apply", op.handle, ctypes.c_void_p(dx.ptr), _lib.SMM_F64, n_src - 1,
                  ctypes.c_void_p(dy.ptr), _lib.SMM_F64, ldy, B, 0.0, 0, None)


@pytest.mark.parametrize("transpose", [True, False])
def test_group_apply_host_pipeline(hip, rng, transpose):
    S, D, L = 642, 130, 4
    ops, csrs = [], []
    imask = (rng.random((L, D)) > 0.3).astype(np.int32)
    frac = rng.random((L, D))
    for l in range(L):
        src, dst, w = random_links(rng, S, D, 400 + 100 * l)
        op = make_op(S, D, src, dst, w)
        op.set_epilogue(imask[l], frac[l])
        ops.append(op)
        csrs.append(op.export_csr())
    grp = OperatorGroup(ops)
    level_index = np.array([2, 0, 3], np.int32)
    masked_levels = np.array([1, 1, 0, 1], np.uint8)
    for n_outer, n_inner, dtype in [(7, 1, np.float64), (5, 2, np.float32), (1, 1, np.float64)]:
        x = field(rng, n_outer * 3 * n_inner, S, dtype=dtype, nan_frac=0.03).reshape(n_outer, 3, n_inner, S)
        ref = oracle.apply_levels(csrs, x, 1, level_index, masked_levels.astype(bool), imask, frac, 0.4,
                                  transpose)
        for chunk in (0, 1, 2, 3):
            y = grp.apply_host(x, level_index, masked_levels, masked=True, remap_area_min=0.4,
                               transpose=transpose, chunk_outer=chunk)
            assert_same(y, ref, exact=True)


# ----------------------------------------------------------------- full-size properties (configs 3 and 5 geometry)

def test_config5_geometry_conserves_area_integral(hip, rng):
    """r1440x721 -> r720x360 conservative at full grid size: the area-weighted integral of every
    batch row is preserved (size-independent property), rows are stochastic."""
    w = gridgen.conservative_weights("r1440x721", "r720x360")
    S, D = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
    assert (S, D) == (1038240, 259200)
    op = make_op(S, D, w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values)
    assert op.n_used_src == S and op.plan_info()["tile_preferred"]
    src, dst = gridgen.parse_grid("r1440x721"), gridgen.parse_grid("r720x360")

    def areas(g):
        return (np.diff(np.sin(np.radians(g.lat_b)))[:, None] * np.radians(np.diff(g.lon_b))[None, :]).ravel()
    x = field(rng, 6, S)
    y = op.apply(to_device(x)).to_host()
    np.testing.assert_allclose((y * areas(dst)).sum(axis=1), (x * areas(src)).sum(axis=1), rtol=1e-11)
    ys = op.apply(to_device(x), flags=_lib.APPLY_KERNEL_SELL).to_host()
    assert_same(y, ys, exact=True)
    assert_same(y[:2], oracle.apply_c(op.export_csr(), x[:2]), exact=True)


def test_config3_geometry_masked_levels(hip, rng):
    """1442x1021 -> r360x180 conservative with two ocean masks (shallow, deep) at full grid size:
    a field that is constant over the ocean regrids to the same constant wherever the unmasked
    fraction reaches remap_area_min and to NaN elsewhere; level sub-selection picks the deep mask."""
    src = gridgen.regular_grid(1442, 1021)
    masks = gridgen.synthetic_ocean_masks(1442, 1021, 2, top=0.66, bottom=0.2)
    w3 = gridgen.ConservativeLevels(src, "r360x180").stack(masks, [5.0, 3000.0])
    ll = w3["link_length"].values
    ops = []
    for i in range(2):
        op = make_op(src.size, 64800, w3["src_address"].values[i, :ll[i]], w3["dst_address"].values[i, :ll[i]],
                     w3["remap_matrix"].values[i, :ll[i], 0])
        imask = op.mask_apply(masks[i])
        op.set_epilogue(imask, w3["dst_grid_frac"].values[i])
        ops.append(op)
    assert max(o.max_row_nnz for o in ops) > 32
    grp = OperatorGroup(ops)
    assert grp.plan_info()["tile_plan"] and grp.plan_info()["slices_per_block"] == 1
    T = 3
    x = np.full((T, 2, 1, src.size), 12.5)
    for i in range(2):
        x[:, i, 0, masks[i] == 0] = np.nan
    y = grp.apply(to_device(x), np.array([0, 1], np.int32), np.array([1, 1], np.uint8), masked=True,
                  remap_area_min=0.5, transpose=True).to_host()          # (T, 1, 2, D)
    frac = w3["dst_grid_frac"].values
    for i in range(2):
        yi = y[:, 0, i, :]
        valid = frac[i] >= 0.5
        assert np.isnan(yi[:, ~valid]).all()
        np.testing.assert_allclose(yi[:, valid], 12.5, rtol=1e-13)
    ysell = grp.apply(to_device(x), np.array([0, 1], np.int32), np.array([1, 1], np.uint8), masked=True,
                      remap_area_min=0.5, transpose=True, flags=_lib.APPLY_KERNEL_SELL).to_host()
    assert_same(y, ysell, exact=True)
    deep_only = grp.apply(to_device(x[:, 1:2]), np.array([1], np.int32), np.array([1, 1], np.uint8),
                          masked=True, remap_area_min=0.5, transpose=True).to_host()
    assert_same(deep_only[:, 0, 0, :], y[:, 0, 1, :], exact=True)


@pytest.mark.parametrize("tag", ["float64", "float32"])
def test_hip_matches_reference_dask_statements(hip, tag):
    """The HIP path against outputs of the reference's own dask statement sequence
    (regrid.py:545-570 run with dask.array, tests/golden/make_dask_golden.py)."""
    z = np.load(os.path.join(GOLDEN, "dask_statements.npz"))
    op = make_op(int(z["n_src"]), int(z["n_dst"]), z["src_address"], z["dst_address"], z["remap_matrix"])
    op.set_epilogue(z["dst_imask"], z["dst_frac"])
    x = z["x_" + tag]
    for masked, amin in ((False, 0.0), (True, 0.0), (True, 0.5), (False, 0.9)):
        ref = z["y_%s_m%d_a%d" % (tag, int(masked), int(amin * 10))]
        for flags in (_lib.APPLY_KERNEL_SELL, 0):
            y = run(op, x.reshape(-1, x.shape[-1]), masked, amin, flags)
            assert_same(y.reshape(ref.shape), ref, rtol=RTOL)          # north-star tolerance
            assert_same(y.reshape(ref.shape), ref, rtol=1e-12)         # and in fact to rounding
    for i in range(3):          # weights.py:47-52 executed with dask.array
        assert np.array_equal(op.mask_apply(z["src_imask_%d" % i]), z["mask_tensordot_%d" % i])
