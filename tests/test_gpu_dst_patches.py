"""Destination patches (SMM_LAYOUT_PATCHES): the device structures follow 4 x 64 patches of a 2-D
target grid, a 4-wave workgroup shares one staged tile.  Results must not depend on the layout:
bit-identical to the row layout and to the oracle (regrid.py:545-570), ragged grid edges included."""
import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import OperatorGroup, SparseOperator, _lib, gridgen, to_device
from smmregrid_amd.weights import compute_weights_matrix, compute_weights_matrix3d
from tests.helpers import assert_same, field, random_links

pytestmark = pytest.mark.gpu


def _ops(w, rng=None):
    args = (w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values, w["dst_address"].values,
            w["remap_matrix"].values)
    dims = w["dst_grid_dims"].values
    return (SparseOperator(*args, device=0, dst_dims=dims, layout="rows"),
            SparseOperator(*args, device=0, dst_dims=dims, layout="patches"))


@pytest.mark.parametrize("src,dst", [("r288x144", "r72x36"),     # nx = 64 + 8, ny a multiple of 4
                                     ("r280x140", "r70x35"),     # ragged in both directions
                                     ("r512x20", "r128x5"),      # two full column blocks, ny = 5
                                     ("r96x48", "r24x12")])      # narrower than one patch
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_patch_layout_is_bit_identical(hip, rng, src, dst, dtype):
    w = gridgen.conservative_weights(src, dst)
    rows, pat = _ops(w)
    assert not rows.plan_info()["dst_patches"] and pat.plan_info()["dst_patches"]
    assert rows.max_row_nnz > 16
    D = rows.n_dst
    imask = (rng.random(D) > 0.2).astype(np.int32)
    frac = rng.random(D)
    for op in (rows, pat):
        op.set_epilogue(imask, frac)
    csr = rows.export_csr()
    for a, b in zip(csr, pat.export_csr()):                    # the canonical CSR does not move
        assert np.array_equal(a, b)
    info = pat.launch_info(40, dtype)
    assert info["kernel"] == "tile" and info["rows_per_block"] == 256     # 4-wave blocks on heavy rows
    for n_batch in (1, 3, 40, 70):
        x = field(rng, n_batch, rows.n_src, dtype=dtype, nan_frac=0.02, inf_frac=0.003)
        for masked, amin in [(False, 0.0), (True, 0.5), (False, 0.9)]:
            ref = oracle.apply_c(csr, x, masked, imask, frac, amin)
            dx = to_device(x)
            for flags in (0, _lib.APPLY_KERNEL_SELL):
                y = pat.apply(dx, masked=masked, remap_area_min=amin, flags=flags).to_host()
                assert_same(y, ref, exact=True)
            assert_same(rows.apply(dx, masked=masked, remap_area_min=amin).to_host(), ref, exact=True)
    # the other entry points see the same operator
    x = field(rng, 9, rows.n_src, nan_frac=0.05)
    ref = oracle.apply_c(csr, x, True, imask, frac, 0.4)
    assert_same(pat.apply_host(x, masked=True, remap_area_min=0.4, chunk_rows=4), ref, exact=True)
    assert_same(pat.apply_sb(to_device(np.ascontiguousarray(x.T)), masked=True, remap_area_min=0.4).to_host(),
                ref, exact=True)
    m = (rng.random(rows.n_src) > 0.4).astype(np.int32)
    assert np.array_equal(pat.mask_apply(m), oracle.mask_apply_c(csr, m))      # weights.py:47-52


def test_patches_on_random_links_and_light_rows(hip, rng):
    """Forced patches work for any row length (they are only *chosen* for 17..48 links): light rows take
    the 4-wave kernels as usual, rows beyond 48 links the single-wave / streamed forms on the slot order."""
    nx, ny, n_src = 130, 9, 5000
    for nnz in (3000, 30000, 90000):                          # ~2.5, ~25 and ~77 links per row
        src, dst, w = random_links(rng, n_src, nx * ny, nnz)
        a = SparseOperator(n_src, nx * ny, src, dst, w, device=0)
        b = SparseOperator(n_src, nx * ny, src, dst, w, device=0, dst_dims=[nx, ny], layout="patches")
        assert b.plan_info()["dst_patches"]
        x = field(rng, 21, n_src, nan_frac=0.02)
        ref = oracle.apply_c(a.export_csr(), x)
        for flags in (0, _lib.APPLY_KERNEL_SELL):
            assert_same(b.apply(to_device(x), flags=flags).to_host(), ref, exact=True)
    with pytest.raises(_lib.SmmError):                         # dims that do not multiply to D
        SparseOperator(n_src, nx * ny, src, dst, w, device=0, dst_dims=[nx, ny + 1], layout="patches")
    with pytest.raises(ValueError):
        SparseOperator(n_src, nx * ny, src, dst, w, device=0, layout="patches")


def test_patch_layout_for_masked_conservative_levels(hip, rng):
    """700 x 350 masked ocean levels -> 2 deg conservative (25 links per row, 180 x 90 target; patches stage
    15 % fewer lines): every level in patch order, the grouped launch matches the oracle.  The automatic
    choice stays rows (patches are opt-in: the coupling of the four waves costs what the traffic saves)."""
    nx, ny, L = 700, 350, 3
    src = gridgen.regular_grid(nx, ny)
    masks = gridgen.synthetic_ocean_masks(nx, ny, L, top=0.7, bottom=0.3)
    w3 = gridgen.ConservativeLevels(src, "r180x90").stack(masks, np.arange(L, dtype=np.float64))
    assert not any(op.plan_info()["dst_patches"] for op in compute_weights_matrix3d(w3, "lev", device=0))
    ops = compute_weights_matrix3d(w3, "lev", device=0, layout="patches")
    assert all(op.plan_info()["dst_patches"] for op in ops)
    csrs = [op.export_csr() for op in ops]
    imask = np.stack([op.mask_apply(masks[i]) for i, op in enumerate(ops)])
    frac = w3["dst_grid_frac"].values
    for i, op in enumerate(ops):
        assert np.array_equal(imask[i], oracle.mask_apply_c(csrs[i], masks[i]))
        op.set_epilogue(imask[i], frac[i])
    grp = OperatorGroup(ops)
    assert grp.launch_info(5, L)["rows_per_block"] == 256
    x = (10.0 + rng.standard_normal((5, L, 1, src.size)))
    x[np.broadcast_to((masks == 0)[None, :, None, :], x.shape)] = np.nan      # land per level
    ml = (~(imask == 1).all(axis=1)).astype(np.uint8)
    for transpose in (True, False):
        y = grp.apply(to_device(x), np.arange(L, dtype=np.int32), ml, masked=True, remap_area_min=0.5,
                      transpose=transpose).to_host()
        ref = oracle.apply_levels(csrs, x, 1, np.arange(L), ml.astype(bool), imask, frac, 0.5, transpose)
        assert_same(y, ref, exact=True)
    # a group cannot mix layouts
    w0 = ops[0]
    other = SparseOperator(w0.n_src, w0.n_dst, w3["src_address"].values[1, :w3["link_length"].values[1]],
                           w3["dst_address"].values[1, :w3["link_length"].values[1]],
                           w3["remap_matrix"].values[1, :w3["link_length"].values[1], 0], device=0,
                           dst_dims=[180, 90], layout="rows")
    with pytest.raises(_lib.SmmError) as e:
        OperatorGroup([ops[0], other])
    assert "layout" in str(e.value)
    w2 = gridgen.conservative_weights(src, "r180x90", src_mask=masks[0])
    assert compute_weights_matrix(w2, device=0, layout="patches").plan_info()["dst_patches"]
