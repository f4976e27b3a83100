"""Reuse plan of the batch-fastest kernel (round 5): a source cell that a later row of the same 16-row tile needs again is
kept in an LDS slot instead of being loaded once per link (smm::build_sb_reuse_codes; the schedule is replayed on the CPU in
tests/cpp/build_harness.cpp).  Same bits with the cache, without it (tuning knob sb_cell_cache = 1), with rounds of 4 loads,
for f32 / f64 fields, both result layouts, the grouped launch, and against the oracle."""
import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import OperatorGroup, SparseOperator, _lib, gridgen, to_device
from tests.helpers import assert_same, field, ragged_links

pytestmark = pytest.mark.gpu


def _shared_column_weights(rng, nx_src, ny_src, nx_dst, ny_dst):
    """Conservative weights between two regular grids whose cell edges do not coincide in longitude: neighbouring
    destination cells share a column of source cells (the structure of BASELINE config 3)."""
    return gridgen.conservative_weights(gridgen.regular_grid(nx_src, ny_src), gridgen.regular_grid(nx_dst, ny_dst))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_shared_cells_through_lds_slots_same_bits(hip, rng, dtype):
    w = _shared_column_weights(rng, 362, 128, 90, 30)            # 4.02 x 4.27 source cells per destination cell
    S, D = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
    op = SparseOperator(S, D, w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values, device=0)
    imask = (rng.random(D) > 0.1).astype(np.int32)
    op.set_epilogue(imask, w["dst_grid_frac"].values)
    csr = op.export_csr()
    # rows of >= 16 links whose last column returns in the next row: the plan finds takes
    rowlen = np.diff(csr[0])
    assert rowlen.max() >= 20
    for B in (1, 2, 7, 130, 257):
        x = field(rng, B, S, dtype=dtype, nan_frac=0.02, inf_frac=0.003)
        ref = oracle.apply_c(csr, x, True, imask, w["dst_grid_frac"].values, 0.5)
        xt = to_device(np.ascontiguousarray(x.T))
        got = {}
        for name, knobs in (("cache", {}), ("plain", dict(sb_cell_cache=1)), ("cache, rounds of 4", dict(sb_loads=4)),
                            ("cache, 5 waves per CU", dict(sb_lds_pad=8192))):
            with _lib.tuning(**knobs):
                got[name] = op.apply_sb(xt, masked=True, remap_area_min=0.5).to_host()
                yk = op.apply_sb(xt, masked=True, remap_area_min=0.5, keep_batch_fastest=True).to_host()
            assert_same(got[name], ref, exact=True)
            assert_same(yk.T, ref, exact=True)
        y32 = op.apply_sb(xt, masked=True, remap_area_min=0.5, out_dtype=np.float32).to_host()      # tiles of 32 rows
        assert_same(y32, ref.astype(np.float32), exact=True)
        # packed X (used cells only): the same plan (cells keep their identity under the renumbering)
        yp = op.apply_sb(to_device(np.ascontiguousarray(x.T[op.used_sources()])), masked=True, remap_area_min=0.5,
                         packed=True).to_host()
        assert_same(yp, ref, exact=True)
    op.close()


def test_grouped_launch_mixes_members_with_and_without_a_plan(hip, rng):
    """A level group whose members differ: conservative levels (plan with takes), a ragged random level and an empty one
    (no plan: they contribute code arrays of zeros to the grouped launch)."""
    w = _shared_column_weights(rng, 181, 64, 45, 15)
    S, D = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
    links = [(w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values[:, 0]),
             ragged_links(rng, S, D, max_len=9),
             (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0)),
             (w["src_address"].values[::-1].copy(), w["dst_address"].values[::-1].copy(), w["remap_matrix"].values[::-1, 0].copy())]
    ops, csrs = [], []
    imask = (rng.random((len(links), D)) > 0.2).astype(np.int32)
    frac = rng.random((len(links), D))
    for i, (s_, d_, v_) in enumerate(links):
        op = SparseOperator(S, D, s_, d_, v_, device=0)
        op.set_epilogue(imask[i], frac[i])
        ops.append(op)
        csrs.append(op.export_csr())
    grp = OperatorGroup(ops)
    ml = np.array([1, 1, 0, 1], np.uint8)
    level_index = np.array([0, 1, 2, 3, 1, 0, 2], np.int32)
    L, B = level_index.size, 70
    x = field(rng, B * L, S, nan_frac=0.02).reshape(B, L, 1, S)
    ref = oracle.apply_levels(csrs, x, 1, level_index, ml.astype(bool), imask, frac, 0.3, True)
    xd = to_device(np.ascontiguousarray(np.transpose(x[:, :, 0, :], (1, 2, 0))))
    for knobs in ({}, dict(sb_cell_cache=1), dict(sb_level_launches=1), dict(sb_loads=4)):
        with _lib.tuning(**knobs):
            y = grp.apply_sb(xd, level_index, ml, masked=True, remap_area_min=0.3).to_host()
        assert_same(y.reshape(ref.shape), ref, exact=True)
    # only members without a plan: the plain grouped kernel
    y = grp.apply_sb(to_device(np.ascontiguousarray(xd.to_host()[[1, 2]])), np.array([1, 2], np.int32), ml, masked=True,
                     remap_area_min=0.3).to_host()
    assert_same(y.reshape(B, 2, D), ref[:, 0, [1, 2]], exact=True)
    grp.close()
    for op in ops:
        op.close()
