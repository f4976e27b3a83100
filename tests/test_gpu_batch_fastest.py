"""The batch-fastest ("SB") operand layout: X (S, B) -- or packed (U, B) -- in, Y (B, D) out,
bit-identical to the oracle (regrid.py:545-570) and to the native-layout kernels."""
import ctypes

import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import SparseOperator, _lib, gridgen, to_device
from smmregrid_amd.device import DeviceArray
from tests.helpers import assert_same, field, ragged_links, random_links

pytestmark = pytest.mark.gpu


def run_sb(op, x_bs, masked=False, amin=0.0, packed=False, out_dtype=np.float64, flags=0):
    """x_bs: host (B, S) as the oracle takes it; the device gets its transpose."""
    xt = np.ascontiguousarray(x_bs.T)
    if packed:
        xt = np.ascontiguousarray(xt[op.used_sources()])
    return op.apply_sb(to_device(xt), masked=masked, remap_area_min=amin, packed=packed, out_dtype=out_dtype,
                       flags=flags).to_host()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("packed", [False, True])
def test_sb_random_matrix(hip, rng, dtype, packed):
    n_src, n_dst = 4096, 1003                      # D not a multiple of the 16-row tile
    src, dst, w = random_links(rng, n_src, n_dst, 6000)
    op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
    csr = op.export_csr()
    imask = (rng.random(n_dst) > 0.2).astype(np.int32)
    frac = rng.random(n_dst)
    op.set_epilogue(imask, frac)
    assert np.array_equal(op.used_sources(), np.unique(csr[1]))
    for n_batch in (1, 2, 3, 5, 64, 127, 128, 129, 300):
        x = field(rng, n_batch, n_src, dtype=dtype, nan_frac=0.02, inf_frac=0.004)
        for masked, amin in [(False, 0.0), (True, 0.0), (True, 0.5), (False, 0.9)]:
            y = run_sb(op, x, masked, amin, packed)
            ref = oracle.apply_c(csr, x, masked, imask, frac, amin)
            assert_same(y, ref, exact=True)


def test_sb_ragged_and_empty_rows(hip, rng):
    n_src, n_dst = 3000, 777
    src, dst, w = ragged_links(rng, n_src, n_dst, max_len=60)     # rows of 0..60 links, many empty
    op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
    x = field(rng, 70, n_src, nan_frac=0.05)
    ref = oracle.apply_c(op.export_csr(), x)
    assert_same(run_sb(op, x), ref, exact=True)
    for knobs in ({"sb_loads": 4}, {"sb_strip": -1}, {"sb_strip": 4}, {"xcd_run": -1}, {"xcd_run": 5}):   # launch-shape knobs: same bits
        with _lib.tuning(**knobs):
            assert_same(run_sb(op, x), ref, exact=True)
    # no links at all: every cell is epilogue(0) = 0
    op0 = SparseOperator(50, 40, np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0), device=0)
    assert (run_sb(op0, field(rng, 5, 50)) == 0.0).all()
    # f32 store is the rounded f64 result
    y32 = run_sb(op, x, out_dtype=np.float32)
    assert y32.dtype == np.float32
    assert_same(y32, ref.astype(np.float32), exact=True)


def test_sb_structured_weights_match_native_layout(hip, rng):
    for method, s, d in [("bil", "r180x90", "r90x45"), ("con", "r144x72", "r36x18"), ("bil", "r96x48", "hp8"),
                         ("nn", "r64x32", "r20x10"), ("con", "r90x45", "r120x60")]:
        w = gridgen.generate_weights(s, d, method=method)
        op = SparseOperator(w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values,
                            w["dst_address"].values, w["remap_matrix"].values, device=0)
        op.set_epilogue(w["dst_grid_imask"].values, w["dst_grid_frac"].values)
        x = field(rng, 37, op.n_src, nan_frac=0.01)
        native = op.apply(to_device(x), remap_area_min=0.5).to_host()
        for packed in (False, True):
            assert_same(run_sb(op, x, False, 0.5, packed), native, exact=True)


def test_sb_row_pitches_and_errors(hip, rng):
    n_src, n_dst, B = 600, 130, 50
    src, dst, w = random_links(rng, n_src, n_dst, 900)
    op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
    x = field(rng, B, n_src)
    ref = oracle.apply_c(op.export_csr(), x)
    ldx, ldy = B + 7, n_dst + 3                      # odd pitches: 16-B loads / stores are element aligned only
    xw = np.full((n_src, ldx), np.nan)
    xw[:, :B] = x.T
    yw = np.full((B, ldy), -7.0)
    dx, dy = to_device(xw), to_device(yw)
    _lib.call("smm_apply_sb", op.handle, ctypes.c_void_p(dx.ptr), _lib.SMM_F64, ldx, ctypes.c_void_p(dy.ptr),
              _lib.SMM_F64, ldy, B, 0.0, 0, None)
    got = dy.to_host()
    assert_same(got[:, :n_dst], ref, exact=True)
    assert (got[:, n_dst:] == -7.0).all()
    with pytest.raises(_lib.SmmError):               # ldx smaller than the batch
        _lib.call("smm_apply_sb", op.handle, ctypes.c_void_p(dx.ptr), _lib.SMM_F64, B - 1, ctypes.c_void_p(dy.ptr),
                  _lib.SMM_F64, ldy, B, 0.0, 0, None)
    with pytest.raises(_lib.SmmError):               # masked without an imask
        op.apply_sb(to_device(np.ascontiguousarray(x.T)), masked=True)
    with pytest.raises(ValueError):
        op.apply_sb(DeviceArray((n_src + 1, B), np.float64))


def test_sb_config2_geometry(hip):
    """r1440x721 -> r360x180 bilinear at full grid size, B = 300 (two full batch tiles + a ragged one):
    sampled rows against the oracle, every row against the native-layout kernel."""
    w = gridgen.bilinear_weights("r1440x721", "r360x180")
    op = SparseOperator(w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values,
                        w["dst_address"].values, w["remap_matrix"].values, device=0)
    op.set_epilogue(w["dst_grid_imask"].values, w["dst_grid_frac"].values)
    B, U = 300, op.n_used_src
    assert U == 259200
    xp = DeviceArray((U, B), np.float64).fill_random(seed=5, mean=250.0, sigma=30.0)     # packed batch-fastest
    xph = xp.to_host()
    xph[::1001, 7] = np.nan
    xp.copy_from_host(xph)
    y = op.apply_sb(xp, remap_area_min=0.5, packed=True).to_host()
    x_bs = np.zeros((B, op.n_src))                   # the same field in the native layout
    x_bs[:, op.used_sources()] = xph.T
    native = op.apply(to_device(x_bs), remap_area_min=0.5).to_host()
    assert_same(y, native, exact=True)
    rows = [0, 7, 127, 128, 255, 256, 299]
    assert_same(y[rows], oracle.apply_c(op.export_csr(), x_bs[rows], False, None, w["dst_grid_frac"].values, 0.5),
                exact=True)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_host_pipeline_packs_the_used_cells(hip, rng, dtype):
    """smm_apply_host on an operator that uses a quarter of its source cells (bilinear 4:1): the staging
    copy packs the used cells batch-fastest and the chunk runs through the batch-fastest kernel -- same
    bits as the whole-row pipeline and as the oracle, for every chunking, pitch and buffer kind."""
    from smmregrid_amd import pinned_empty
    w = gridgen.bilinear_weights("r360x180", "r90x45")
    op = SparseOperator(w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values,
                        w["dst_address"].values, w["remap_matrix"].values, device=0)
    assert 2 * op.n_used_src <= op.n_src
    imask = (rng.random(op.n_dst) > 0.2).astype(np.int32)
    op.set_epilogue(imask, w["dst_grid_frac"].values)
    B = 150
    x = field(rng, B, op.n_src, dtype=dtype, nan_frac=0.02, inf_frac=0.003)
    ref = oracle.apply_c(op.export_csr(), x, True, imask, w["dst_grid_frac"].values, 0.5)
    for chunk in (0, 7, 40, 128, 149):
        y = op.apply_host(x, masked=True, remap_area_min=0.5, chunk_rows=chunk)
        assert_same(y, ref, exact=True)
    y = op.apply_host(x, masked=True, remap_area_min=0.5, flags=_lib.APPLY_HOST_NO_PACK)
    assert_same(y, ref, exact=True)
    wide = np.zeros((B, op.n_src + 5), dtype=dtype)              # ldx > S
    wide[:, :op.n_src] = x
    assert_same(op.apply_host(wide[:, :op.n_src], masked=True, remap_area_min=0.5, chunk_rows=33), ref, exact=True)
    xp = pinned_empty(x.shape, dtype)
    xp[...] = x
    yp = pinned_empty((B, op.n_dst), np.float64)
    op.apply_host(xp, out=yp, masked=True, remap_area_min=0.5)
    assert_same(np.array(yp), ref, exact=True)
    # short batches: packed from 8 rows on (round 6; 32 before), whole rows below
    for rows in (20, 9, 8, 7, 1):
        assert_same(op.apply_host(x[:rows], masked=True, remap_area_min=0.5), ref[:rows], exact=True)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("transpose", [True, False])
def test_group_apply_sb_matches_the_oracle(hip, rng, transpose, dtype):
    """Masked levels with the field kept batch-fastest per level, X (L, S, B): regrid.py:387-427 through the
    batch-fastest kernel (one grouped launch) -- level sub-selection and repeats, per-level masks, odd batch sizes,
    f32 fields (promoted to f64 as regrid.py:550 does) and the opt-in f32 store (the rounded f64 result)."""
    from smmregrid_amd import OperatorGroup
    S, D, n_ops = 900, 217, 4
    ops, csrs = [], []
    imask = (rng.random((n_ops, D)) > 0.3).astype(np.int32)
    frac = rng.random((n_ops, D))
    for i in range(n_ops):
        src, dst, w = (random_links(rng, S, D, 1500 + 300 * i) if i % 2 else ragged_links(rng, S, D, max_len=25))
        op = SparseOperator(S, D, src, dst, w, device=0)
        op.set_epilogue(imask[i], frac[i])
        ops.append(op)
        csrs.append(op.export_csr())
    grp = OperatorGroup(ops)
    masked_levels = np.array([1, 0, 1, 1], np.uint8)
    for level_index, B in [([0, 1, 2, 3], 130), ([2, 0, 2], 7), ([3], 1), ([1, 1, 0, 3, 2], 64)]:
        L = len(level_index)
        x = field(rng, B * L, S, dtype=dtype, nan_frac=0.03).reshape(B, L, 1, S)   # native layout for the oracle
        ref = oracle.apply_levels(csrs, x, 1, np.asarray(level_index), masked_levels.astype(bool), imask, frac,
                                  0.4, transpose)                            # (B, 1, L, D) / (L, B, 1, D)
        x_sb = np.ascontiguousarray(np.transpose(x[:, :, 0, :], (1, 2, 0)))   # (L, S, B)
        y = grp.apply_sb(to_device(x_sb), level_index, masked_levels, masked=True, remap_area_min=0.4,
                         transpose=transpose).to_host()
        assert y.dtype == np.float64
        assert_same(y.reshape(ref.shape), ref, exact=True)
        y32 = grp.apply_sb(to_device(x_sb), level_index, masked_levels, masked=True, remap_area_min=0.4,
                           transpose=transpose, out_dtype=np.float32).to_host()
        assert y32.dtype == np.float32
        assert_same(y32.reshape(ref.shape), ref.astype(np.float32), exact=True)
        native = grp.apply(to_device(x), level_index, masked_levels, masked=True, remap_area_min=0.4,
                           transpose=transpose).to_host()
        assert_same(y.reshape(native.shape), native, exact=True)
    with pytest.raises(_lib.SmmError):                                        # packed is per operator
        grp.apply_sb(to_device(np.zeros((1, S, 4))), [0], flags=_lib.APPLY_SB_PACKED)


@pytest.mark.parametrize("transpose", [True, False])
def test_group_host_pipeline_packs_per_level(hip, rng, transpose):
    """smm_group_apply_host when the selected levels use at most half of their source cells (ocean levels
    thinning out with depth): each level's used cells of a chunk are packed batch-fastest and the level
    runs through the batch-fastest kernel -- same bits as the whole-row pipeline and the oracle."""
    from smmregrid_amd import OperatorGroup
    S, D, n_ops = 4000, 230, 4
    ops, csrs = [], []
    imask = (rng.random((n_ops, D)) > 0.3).astype(np.int32)
    frac = rng.random((n_ops, D))
    for i in range(n_ops):
        src, dst, w = random_links(rng, S, D, 300 + 250 * i)             # U_i well below S / 2
        op = SparseOperator(S, D, src, dst, w, device=0)
        op.set_epilogue(imask[i], frac[i])
        ops.append(op)
        csrs.append(op.export_csr())
    assert sum(op.n_used_src for op in ops) * 2 <= n_ops * S
    grp = OperatorGroup(ops)
    masked_levels = np.array([1, 0, 1, 1], np.uint8)
    for level_index, n_outer, n_inner, dtype in [([0, 1, 2, 3], 40, 1, np.float64), ([2, 0, 2], 9, 5, np.float32),
                                                  ([3], 33, 1, np.float64)]:
        L = len(level_index)
        x = field(rng, n_outer * L * n_inner, S, dtype=dtype, nan_frac=0.03).reshape(n_outer, L, n_inner, S)
        ref = oracle.apply_levels(csrs, x, 1, np.asarray(level_index), masked_levels.astype(bool), imask, frac, 0.4,
                                  transpose)
        for chunk in (0, 1, 4, 7, 33):          # small explicit chunks take the whole-row pipeline
            y = grp.apply_host(x, level_index, masked_levels, masked=True, remap_area_min=0.4, transpose=transpose,
                               chunk_outer=chunk)
            assert_same(y, ref, exact=True)
        y = grp.apply_host(x, level_index, masked_levels, masked=True, remap_area_min=0.4, transpose=transpose,
                           flags=_lib.APPLY_HOST_NO_PACK)
        assert_same(y, ref, exact=True)


# ------------------------------------------------------------------ results kept batch-fastest (SMM_APPLY_SB_Y_SB)

@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sb_result_kept_batch_fastest(hip, rng, dtype):
    """Y (D, B): the transpose of what smm_apply_sb / smm_apply write, bit for bit -- every batch size
    around the 128-entry tile and the odd tails, packed and whole fields, f32 store, row pitches."""
    n_src, n_dst = 3000, 517
    src, dst, w = random_links(rng, n_src, n_dst, 5000)
    op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
    csr = op.export_csr()
    imask = (rng.random(n_dst) > 0.2).astype(np.int32)
    frac = rng.random(n_dst)
    op.set_epilogue(imask, frac)
    for n_batch in (1, 2, 3, 64, 127, 128, 129, 301):
        x = field(rng, n_batch, n_src, dtype=dtype, nan_frac=0.02, inf_frac=0.004)
        xt = np.ascontiguousarray(x.T)
        for masked, amin in [(False, 0.0), (True, 0.5)]:
            ref = oracle.apply_c(csr, x, masked, imask, frac, amin)
            y = op.apply_sb(to_device(xt), masked=masked, remap_area_min=amin, keep_batch_fastest=True)
            assert y.shape == (n_dst, n_batch) and y.layout == "sb"
            assert_same(y.to_host().T, ref, exact=True)
        yp = op.apply_sb(to_device(np.ascontiguousarray(xt[op.used_sources()])), packed=True, keep_batch_fastest=True)
        assert_same(yp.to_host().T, oracle.apply_c(csr, x), exact=True)
        y32 = op.apply_sb(to_device(xt), out_dtype=np.float32, keep_batch_fastest=True).to_host()
        assert y32.dtype == np.float32
        assert_same(y32.T, oracle.apply_c(csr, x).astype(np.float32), exact=True)
    # odd pitches on both sides: 16-B accesses are element aligned only; the padding stays untouched
    B, ldx, ldy = 50, 57, 53
    x = field(rng, B, n_src, dtype=dtype)
    xw = np.full((n_src, ldx), np.nan, dtype=dtype)
    xw[:, :B] = x.T
    yw = np.full((n_dst, ldy), -7.0)
    dx, dy = to_device(xw), to_device(yw)
    _lib.call("smm_apply_sb", op.handle, ctypes.c_void_p(dx.ptr), _lib.SMM_F64 if dtype == np.float64 else _lib.SMM_F32,
              ldx, ctypes.c_void_p(dy.ptr), _lib.SMM_F64, ldy, B, 0.0, _lib.APPLY_SB_Y_SB, None)
    got = dy.to_host()
    assert_same(got[:, :B].T, oracle.apply_c(csr, x), exact=True)
    assert (got[:, B:] == -7.0).all()
    with pytest.raises(_lib.SmmError):               # ldy must hold a batch now, not a destination row
        _lib.call("smm_apply_sb", op.handle, ctypes.c_void_p(dx.ptr), _lib.SMM_F64, ldx, ctypes.c_void_p(dy.ptr),
                  _lib.SMM_F64, B - 1, B, 0.0, _lib.APPLY_SB_Y_SB, None)


def test_chain_of_regrids_stays_batch_fastest(hip, rng):
    """r144x72 -> r72x36 -> r24x12 on a device-resident batch-fastest field: the first result feeds the
    second operator as it is, and `SparseOperator.apply` routes by the layout tag."""
    w1 = gridgen.generate_weights("r144x72", "r72x36", method="con")
    w2 = gridgen.generate_weights("r72x36", "r24x12", method="bil")
    ops = []
    for w in (w1, w2):
        op = SparseOperator(w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values,
                            w["dst_address"].values, w["remap_matrix"].values, device=0)
        op.set_epilogue(w["dst_grid_imask"].values, w["dst_grid_frac"].values)
        ops.append(op)
    x = field(rng, 45, ops[0].n_src, nan_frac=0.01)
    mid = oracle.apply_c(ops[0].export_csr(), x, False, None, w1["dst_grid_frac"].values, 0.5)
    ref = oracle.apply_c(ops[1].export_csr(), mid, False, None, w2["dst_grid_frac"].values, 0.5)
    x_sb = to_device(np.ascontiguousarray(x.T), layout="sb")
    y1 = ops[0].apply(x_sb, remap_area_min=0.5, keep_batch_fastest=True)          # routed by the tag
    assert y1.layout == "sb" and y1.shape == (ops[0].n_dst, 45)
    y2 = ops[1].apply(y1, remap_area_min=0.5)                                     # (B, D) out at the end
    assert y2.layout == "bs" and y2.shape == (45, ops[1].n_dst)
    assert_same(y2.to_host(), ref, exact=True)
    with pytest.raises(ValueError):                  # a native-layout field cannot keep a layout it is not in
        ops[0].apply(to_device(x), keep_batch_fastest=True)


def test_group_result_kept_batch_fastest(hip, rng):
    S, D, n_ops, B = 2500, 300, 3, 70
    ops, csrs = [], []
    imask = (rng.random((n_ops, D)) > 0.3).astype(np.int32)
    frac = rng.random((n_ops, D))
    for i in range(n_ops):
        src, dst, w = random_links(rng, S, D, 2000 + 500 * i)
        op = SparseOperator(S, D, src, dst, w, device=0)
        op.set_epilogue(imask[i], frac[i])
        ops.append(op)
        csrs.append(op.export_csr())
    from smmregrid_amd import OperatorGroup
    grp = OperatorGroup(ops)
    level_index = [2, 0, 1, 2]
    masked_levels = np.array([1, 0, 1], np.uint8)
    x = field(rng, B * 4, S, nan_frac=0.03).reshape(B, 4, S)                       # (B, L, S) for the oracle
    ref = oracle.apply_levels(csrs, x, 1, np.asarray(level_index), masked_levels.astype(bool), imask, frac, 0.4, True)
    x_sb = to_device(np.ascontiguousarray(x.transpose(1, 2, 0)), layout="sb")      # (L, S, B)
    y = grp.apply_sb(x_sb, level_index, masked_levels, masked=True, remap_area_min=0.4, keep_batch_fastest=True)
    assert y.shape == (4, D, B) and y.layout == "sb"
    assert_same(y.to_host().transpose(2, 0, 1), ref, exact=True)                   # (B, L, D)


# ------------------------------------------------------------------ conservative operators, padded pitches

def _con_operator(src, dst, mask=None):
    w = gridgen.generate_weights(src, dst, method="con", src_mask=mask)
    op = SparseOperator(w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values,
                        w["dst_address"].values, w["remap_matrix"].values, device=0)
    return w, op


@pytest.mark.parametrize("src,dst", [("r144x72", "r36x18"), ("r360x180", "r72x36"), ("r720x360", "r120x60"),
                                     ("r200x100", "r90x45")])
def test_sb_conservative_masked_operators(hip, rng, src, dst):
    """Conservative operators with a source mask (cells shared between links and rows, ragged rows at the
    coasts) in the batch-fastest layout: batch sizes around the tile and its tails, masks, thresholds,
    non-finite values, f32 stores, results kept batch-fastest."""
    g = gridgen.parse_grid(src)
    mask = (rng.random(g.size) > 0.3).astype(np.int32)
    w, op = _con_operator(src, dst, mask)
    imask = op.mask_apply(mask)
    frac = w["dst_grid_frac"].values
    op.set_epilogue(imask, frac)
    csr = op.export_csr()
    for n_batch in (2, 3, 15, 16, 17, 33, 120, 257):
        x = field(rng, n_batch, op.n_src, nan_frac=0.05, inf_frac=0.003)
        x[:, mask == 0] = np.nan
        for masked, amin in [(False, 0.0), (True, 0.5)]:
            assert_same(run_sb(op, x, masked, amin), oracle.apply_c(csr, x, masked, imask, frac, amin), exact=True)
        xt = to_device(np.ascontiguousarray(x.T))
        ref = oracle.apply_c(csr, x, True, imask, frac, 0.5)
        y = op.apply_sb(xt, masked=True, remap_area_min=0.5, keep_batch_fastest=True)
        assert_same(y.to_host().T, ref, exact=True)
        y32 = op.apply_sb(xt, masked=True, remap_area_min=0.5, out_dtype=np.float32).to_host()
        assert_same(y32, ref.astype(np.float32), exact=True)
    xf = field(rng, 40, op.n_src)
    yf = op.apply_sb(to_device(np.ascontiguousarray(xf.T)), flags=_lib.APPLY_NO_FILL).to_host()
    assert_same(yf, oracle.apply_c(csr, xf), exact=True)


def test_sb_padded_pitch_through_n_batch(hip, rng):
    """`n_batch` with a padded pitch: cells that start on 128-B lines (pitch a multiple of 16 doubles)."""
    w, op = _con_operator("r144x72", "r36x18")
    op.set_epilogue(w["dst_grid_imask"].values, w["dst_grid_frac"].values)
    csr = op.export_csr()
    for B, ldx in [(120, 128), (50, 57), (33, 48)]:
        x = field(rng, B, op.n_src, nan_frac=0.02)
        xw = np.full((op.n_src, ldx), np.nan)
        xw[:, :B] = x.T
        ref = oracle.apply_c(csr, x, False, None, w["dst_grid_frac"].values, 0.5)
        y = op.apply_sb(to_device(xw), remap_area_min=0.5, n_batch=B)
        assert y.shape == (B, op.n_dst)
        assert_same(y.to_host(), ref, exact=True)
        yk = op.apply_sb(to_device(xw), remap_area_min=0.5, n_batch=B, keep_batch_fastest=True)
        assert yk.shape == (op.n_dst, B)
        assert_same(yk.to_host().T, ref, exact=True)
    with pytest.raises(ValueError):
        op.apply_sb(to_device(np.zeros((op.n_src, 8))), n_batch=9)


def test_group_sb_with_padded_pitch(hip, rng):
    """Masked levels kept batch-fastest per level with a pitch on 128-B lines, (L, S, 128) for T = 120:
    (T, L, D) and (L, D, T) results."""
    from smmregrid_amd import OperatorGroup
    from smmregrid_amd.weights import compute_weights_matrix3d
    nx, ny, n_lev, T = 288, 144, 3, 120
    srcg = gridgen.regular_grid(nx, ny)
    masks = gridgen.synthetic_ocean_masks(nx, ny, n_lev, top=0.7, bottom=0.2)
    w3 = gridgen.ConservativeLevels(srcg, "r72x36").stack(masks, np.arange(n_lev, dtype=float))
    ops = compute_weights_matrix3d(w3, "lev", device=0)
    imask = np.stack([op.mask_apply(masks[i]) for i, op in enumerate(ops)])
    frac = w3["dst_grid_frac"].values
    for i, op in enumerate(ops):
        op.set_epilogue(imask[i], frac[i])
    grp = OperatorGroup(ops)
    x = 5.0 + rng.standard_normal((T, n_lev, nx * ny))
    for lv in range(n_lev):
        x[:, lv, masks[lv] == 0] = np.nan
    ml = np.ones(n_lev, np.uint8)
    ref = oracle.apply_levels([op.export_csr() for op in ops], x, 1, np.arange(n_lev), ml.astype(bool), imask, frac, 0.5, True)
    xp = np.full((n_lev, nx * ny, 128), np.nan)
    xp[:, :, :T] = x.transpose(1, 2, 0)
    dxp = to_device(xp)
    y = grp.apply_sb(dxp, np.arange(n_lev, dtype=np.int32), ml, masked=True, remap_area_min=0.5, n_batch=T)
    assert y.shape == (T, n_lev, ops[0].n_dst)
    assert_same(y.to_host(), ref, exact=True)
    yk = grp.apply_sb(dxp, np.arange(n_lev, dtype=np.int32), ml, masked=True, remap_area_min=0.5, n_batch=T,
                      keep_batch_fastest=True)
    assert yk.shape == (n_lev, ops[0].n_dst, T)
    assert_same(yk.to_host().transpose(2, 0, 1), ref, exact=True)
