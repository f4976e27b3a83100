"""Round 4: launch grids beyond the limit are split (smm_debug_set_grid_limit forces the path on small
inputs), operator creation is threaded (smm_set_host_threads: same operator whatever the count), and the
LDS-DMA staging falls back to register staging for fields that are not 16-B aligned."""
import ctypes

import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import OperatorGroup, SparseOperator, _lib, gridgen, to_device
from smmregrid_amd.device import DeviceArray
from smmregrid_amd.weights import compute_weights_matrix3d
from tests.helpers import assert_same, field, ragged_links, random_links

pytestmark = pytest.mark.gpu
T, S_ = _lib.APPLY_KERNEL_TILE, _lib.APPLY_KERNEL_SELL


@pytest.fixture
def grid_limit():
    """Lowers the launch-grid limit for one test (explicit debug setter, no environment variable)."""
    def setter(n):
        _lib.call("smm_debug_set_grid_limit", int(n))
    yield setter
    _lib.call("smm_debug_set_grid_limit", 0)


def _operator(w, device=0):
    op = SparseOperator(w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values,
                        w["dst_address"].values, w["remap_matrix"].values, device=device)
    return op


@pytest.mark.parametrize("method,src,dst", [("bil", "r360x180", "r90x45"), ("con", "r360x180", "r120x60")])
def test_grids_beyond_the_limit_are_launched_in_parts(hip, rng, grid_limit, method, src, dst):
    """smm_apply (tile and SELL kernels) and smm_apply_sb with the limit forced far below the grid the batch
    needs: the parts together give the bits of the single launch and of the oracle; a limit no single batch
    row fits is an error, not a partial result."""
    w = gridgen.generate_weights(src, dst, method=method)
    op = _operator(w)
    imask = (rng.random(op.n_dst) > 0.1).astype(np.int32)
    op.set_epilogue(imask, w["dst_grid_frac"].values)
    x = field(rng, 37, op.n_src, nan_frac=0.02)
    ref = oracle.apply_c(op.export_csr(), x, True, imask, w["dst_grid_frac"].values, 0.5)
    xd = to_device(x)
    xsb = to_device(np.ascontiguousarray(x.T))
    whole = {fl: op.apply(xd, masked=True, remap_area_min=0.5, flags=fl).to_host() for fl in (T, S_)}
    for fl in (T, S_):
        info = op.launch_info(37, np.float64, flags=fl)
        assert_same(whole[fl], ref, exact=True)
        for limit in (info["n_blocks"] - 1, max(info["n_blocks"] // 7, 64), 64):
            grid_limit(limit)
            y = op.apply(xd, masked=True, remap_area_min=0.5, flags=fl).to_host()
            assert_same(y, ref, exact=True)
        grid_limit(1)
        with pytest.raises(_lib.SmmError):
            op.apply(xd, masked=True, remap_area_min=0.5, flags=fl)
        grid_limit(0)
    n_dtiles = -(-op.n_dst // 16)
    for keep in (False, True):
        for limit in (n_dtiles, 3 * n_dtiles + 5):     # one batch tile of 128 entries per launch / three
            grid_limit(limit)
            xl = DeviceArray((op.n_src, 300), np.float64)
            big = field(rng, 300, op.n_src, nan_frac=0.01)
            xl.copy_from_host(np.ascontiguousarray(big.T))
            y = op.apply_sb(xl, masked=True, remap_area_min=0.5, keep_batch_fastest=keep).to_host()
            r2 = oracle.apply_c(op.export_csr(), big, True, imask, w["dst_grid_frac"].values, 0.5)
            assert_same(y.T if keep else y, r2, exact=True)
        grid_limit(n_dtiles - 1)
        with pytest.raises(_lib.SmmError):
            op.apply_sb(xsb, masked=True, remap_area_min=0.5, keep_batch_fastest=keep)
        grid_limit(0)


def test_group_launch_is_split_over_outer_and_inner_ranges(hip, rng, grid_limit):
    """smm_group_apply over (outer, level, inner) batches: the split walks halves of the outer range, then
    of the inner range (strides carry the offsets), both output orders."""
    nx, ny, n_lev = 96, 48, 5
    masks = gridgen.synthetic_ocean_masks(nx, ny, n_lev)
    w3 = gridgen.ConservativeLevels(gridgen.regular_grid(nx, ny), "r24x12").stack(masks, np.arange(n_lev, dtype=float))
    ops = compute_weights_matrix3d(w3, "lev", device=0)
    dmask = np.stack([op.mask_apply(masks[i]) for i, op in enumerate(ops)])
    for i, op in enumerate(ops):
        op.set_epilogue(dmask[i], w3["dst_grid_frac"].values[i])
    group = OperatorGroup(ops)
    n_outer, n_inner, S, D = 6, 5, nx * ny, 24 * 12
    x = 10.0 + rng.standard_normal((n_outer, n_lev, n_inner, S))
    x[rng.random(x.shape) < 0.01] = np.nan
    lev = np.arange(n_lev, dtype=np.int32)
    ml = (~(dmask == 1).all(axis=1)).astype(np.uint8)
    xd = to_device(x)
    csrs = [op.export_csr() for op in ops]
    for transpose in (True, False):
        ref = oracle.apply_levels(csrs, x, 1, lev, ml, dmask, w3["dst_grid_frac"].values, 0.5, transpose)
        whole = group.apply(xd, lev, ml, masked=True, remap_area_min=0.5, transpose=transpose).to_host()
        assert_same(whole, ref, exact=True)
        n_blocks = group.launch_info(n_outer, n_lev, n_inner)["n_blocks"]
        for limit in (n_blocks - 1, n_blocks // 3, max(n_blocks // 11, 16)):
            grid_limit(limit)
            y = group.apply(xd, lev, ml, masked=True, remap_area_min=0.5, transpose=transpose).to_host()
            assert_same(y, ref, exact=True)
            grid_limit(0)
    group.close()
    for op in ops:
        op.close()


@pytest.mark.parametrize("kind", ["random", "ragged", "cdo_order"])
def test_operator_does_not_depend_on_the_builder_threads(hip, rng, kind):
    """smm_set_host_threads(1) and (5): identical CSR, identical kernel plan, identical results -- for link
    lists in random order and in CDO's (dst, src) order (the builder's sort-free path)."""
    n_src, n_dst = 9000, 2100
    if kind == "ragged":
        src, dst, w = ragged_links(rng, n_src, n_dst, max_len=45)
    else:
        src, dst, w = random_links(rng, n_src, n_dst, 30000, dup_frac=0.2)
        if kind == "cdo_order":
            o = np.lexsort((src, dst))
            src, dst, w = src[o], dst[o], w[o]
    x = to_device(field(rng, 9, n_src, nan_frac=0.01))
    prev = ctypes.c_int(-1)
    got = []
    try:
        for nt in (1, 5):
            _lib.call("smm_set_host_threads", nt, ctypes.byref(prev))
            op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
            got.append((op.export_csr(), op.plan_info(), op.apply(x).to_host(), op.apply(x, flags=S_).to_host()))
            assert op.create_ms > 0
            op.close()
    finally:
        _lib.call("smm_set_host_threads", 0, None)
    (c1, p1, y1, s1), (c5, p5, y5, s5) = got
    assert all(np.array_equal(a, b) for a, b in zip(c1, c5)) and p1 == p5
    assert_same(y5, y1, exact=True)
    assert_same(s5, s1, exact=True)
    rowptr, col, val = oracle.coo_to_csr_c(n_src, n_dst, src, dst, w)
    assert np.array_equal(c5[0], rowptr) and np.array_equal(c5[1], col)
    assert np.array_equal(c5[2].view(np.uint64), val.view(np.uint64))


def test_levels_created_by_worker_threads_match_serial_creation(hip, rng):
    """compute_weights_matrix3d creates the levels on a pool of host threads: same operators, in level order."""
    nx, ny, n_lev = 120, 60, 9
    masks = gridgen.synthetic_ocean_masks(nx, ny, n_lev)
    w3 = gridgen.ConservativeLevels(gridgen.regular_grid(nx, ny), "r36x18").stack(masks, np.arange(n_lev, dtype=float))
    serial = compute_weights_matrix3d(w3, "lev", device=0, workers=1)
    pooled = compute_weights_matrix3d(w3, "lev", device=0)
    for a, b in zip(serial, pooled):
        assert all(np.array_equal(u, v) for u, v in zip(a.export_csr(), b.export_csr()))
        assert a.plan_info() == b.plan_info()
    for op in serial + pooled:
        op.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_dma_request_on_unaligned_fields_falls_back_bit_equal(hip, rng, dtype):
    """The tuning knob tile_staging = 2 asks for LDS-DMA staging, which moves aligned 16-B pieces.  A field whose base or row pitch
    is not a multiple of 16 B must take the register-staged kernel (the DMA kernel skips the piece rotation
    such rows need): launch_info -- which assumes an aligned field -- says tile-dma, the results of the odd
    base / odd pitch fields equal the oracle bit for bit."""
    w = gridgen.generate_weights("r360x180", "r90x45", method="bil")
    op = _operator(w)
    op.set_epilogue(None, w["dst_grid_frac"].values)
    with _lib.tuning(tile_staging=_lib.STAGING_DMA, tile_rows_per_step=1):
        assert op.launch_info(12, dtype, flags=T)["kernel"] == "tile-dma"
    S = op.n_src
    n_batch = 12
    for pitch, offset in ((S + 1, 0), (S, 1), (S + 3, 1), (S + (16 // np.dtype(dtype).itemsize), 0)):
        buf = DeviceArray((n_batch * pitch + 4,), dtype)
        host = np.zeros(n_batch * pitch + 4, dtype=dtype)
        x = field(rng, n_batch, S, dtype=dtype, nan_frac=0.02)
        view = host[offset:offset + n_batch * pitch].reshape(n_batch, pitch)
        view[:, :S] = x
        buf.copy_from_host(host)
        xv = DeviceArray((n_batch, pitch), dtype, ptr=buf.ptr + offset * np.dtype(dtype).itemsize, base=buf)
        ref = oracle.apply_c(op.export_csr(), x, False, None, w["dst_grid_frac"].values, 0.5)
        for knobs in (dict(tile_staging=_lib.STAGING_DMA, tile_rows_per_step=1),
                      dict(tile_staging=_lib.STAGING_DMA, tile_rows_per_step=2), {}):
            with _lib.tuning(**knobs):
                y = op.apply(xv, remap_area_min=0.5, flags=T).to_host()
            assert_same(y, ref, exact=True)
        buf.free()


def test_group_sb_levels_in_one_grouped_launch(hip, rng):
    """smm_group_apply_sb runs every data level in ONE grouped launch (round 5: the levels' pointers travel in the
    kernel arguments; round 4's stream pool of per-level launches is gone); the tuning knob sb_level_launches selects
    one launch per level on the caller's stream.  Same bits either way; on a caller stream of its own the call is
    ordered between the upload queued before it and the download queued after it.  More data levels than one launch's
    arguments hold (88) -- and a forced launch-grid limit -- cut the grouped form into several launches with the
    same bits."""
    from smmregrid_amd.device import Stream
    S, D, n_ops, B = 1100, 260, 7, 50
    ops, csrs = [], []
    imask = (rng.random((n_ops, D)) > 0.25).astype(np.int32)
    frac = rng.random((n_ops, D))
    for i in range(n_ops):
        src, dst, w = (random_links(rng, S, D, 2500 + 200 * i) if i % 2 else ragged_links(rng, S, D, max_len=30))
        op = SparseOperator(S, D, src, dst, w, device=0)
        op.set_epilogue(imask[i], frac[i])
        ops.append(op)
        csrs.append(op.export_csr())
    grp = OperatorGroup(ops)
    level_index = np.array([0, 1, 2, 3, 4, 5, 6, 3, 1, 0, 6, 5, 2, 4, 0, 1, 2, 3, 4], dtype=np.int32)   # 19 data levels
    ml = np.array([1, 0, 1, 1, 0, 1, 1], np.uint8)
    L = level_index.size
    x = field(rng, B * L, S, nan_frac=0.02).reshape(B, L, 1, S)
    ref = oracle.apply_levels(csrs, x, 1, level_index, ml.astype(bool), imask, frac, 0.4, True)   # (B, 1, L, D)
    x_sb = np.ascontiguousarray(np.transpose(x[:, :, 0, :], (1, 2, 0)))                            # (L, S, B)
    xd = to_device(x_sb)
    for per_level in (0, 1):
        for _ in range(2):
            with _lib.tuning(sb_level_launches=per_level):
                y = grp.apply_sb(xd, level_index, ml, masked=True, remap_area_min=0.4).to_host()
            assert_same(y.reshape(ref.shape), ref, exact=True)
    # 200 data levels cycling through the 7 members: three grouped launches (88 + 88 + 24); then a launch-grid limit
    # that lets only a few levels into one launch
    many = np.tile(level_index, 11)[:200].astype(np.int32)
    xm = np.ascontiguousarray(np.tile(x_sb, (11, 1, 1))[:200])
    xmd = to_device(xm)
    ym = grp.apply_sb(xmd, many, ml, masked=True, remap_area_min=0.4).to_host()
    for rep in range(11):
        n = min(L, 200 - rep * L)
        assert_same(ym[:, rep * L:rep * L + n], ref[:, 0, :n], exact=True)
    per_level = -(-D // 16) * -(-B // 128)
    for limit in (per_level, 3 * per_level + 1):
        _lib.call("smm_debug_set_grid_limit", limit)
        try:
            y2 = grp.apply_sb(xmd, many, ml, masked=True, remap_area_min=0.4).to_host()
        finally:
            _lib.call("smm_debug_set_grid_limit", 0)
        assert np.array_equal(y2.view(np.uint64), ym.view(np.uint64))
    # kept batch-fastest (Y (L, D, B)) through the grouped launch against one launch per level
    yk = grp.apply_sb(xd, level_index, ml, masked=True, remap_area_min=0.4, keep_batch_fastest=True).to_host()
    with _lib.tuning(sb_level_launches=1):
        yk4 = grp.apply_sb(xd, level_index, ml, masked=True, remap_area_min=0.4, keep_batch_fastest=True).to_host()
    assert np.array_equal(yk.view(np.uint64), yk4.view(np.uint64))
    # a caller stream of its own: H2D, apply, D2H all queued on it, one synchronisation at the end
    st = Stream()
    xs = DeviceArray(x_sb.shape, np.float64)
    ys = DeviceArray((B, L, D), np.float64)
    for rep in range(3):
        x2 = field(rng, B * L, S, nan_frac=0.02).reshape(B, L, 1, S)
        ref2 = oracle.apply_levels(csrs, x2, 1, level_index, ml.astype(bool), imask, frac, 0.4, True)
        xs.copy_from_host(np.ascontiguousarray(np.transpose(x2[:, :, 0, :], (1, 2, 0))), stream=st)
        grp.apply_sb(xs, level_index, ml, y=ys, masked=True, remap_area_min=0.4, stream=st)
        got = ys.to_host(stream=st)
        assert_same(got.reshape(ref2.shape), ref2, exact=True)
    st.close()
    grp.close()
    for op in ops:
        op.close()


def test_a_failing_level_fails_the_pooled_creation_cleanly(hip, rng):
    """One level with an address outside the source grid: compute_weights_matrix3d raises the library's
    error (lowest bad link index in the message) whatever worker hits it, and the operators the other
    workers made are released (creating the same levels again afterwards works)."""
    nx, ny, n_lev = 60, 30, 6
    masks = gridgen.synthetic_ocean_masks(nx, ny, n_lev)
    w3 = gridgen.ConservativeLevels(gridgen.regular_grid(nx, ny), "r20x10").stack(masks, np.arange(n_lev, dtype=float))
    good = w3["src_address"].values.copy()
    bad = good.copy()
    bad[3, 5] = nx * ny + 7
    bad[3, 9] = 0
    dims = w3["src_address"].dims
    w3["src_address"] = (dims, bad)
    with pytest.raises(_lib.SmmError) as err:
        compute_weights_matrix3d(w3, "lev", device=0)
    assert "src_address[5]" in str(err.value)
    w3["src_address"] = (dims, good)
    ops = compute_weights_matrix3d(w3, "lev", device=0, workers=3)
    assert len(ops) == n_lev and all(op.nnz > 0 for op in ops)
    for op in ops:
        op.close()
