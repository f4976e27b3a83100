"""GPU parity at the BASELINE.json geometries with the launch shapes the bench times.

The small-batch parity tests never reach the production walk lengths (4 / 64 / 128 batch rows
per workgroup, several batch tiles per destination block, the ragged last tile), the multi-row
steps of small tiles, the `big_operator` walk of config 4 or the 75-level map of config 3.
These tests do: fields are generated on the device (a few rows get NaN / inf poked in from the
host), sampled batch rows are compared bit for bit with the CPU oracle
(regrid.py:536-570, :387-418) and every row is compared bit for bit between the LDS-tile kernel
and the independent SELL kernel.

The oracle is fed ITS OWN matrix: `oracle.coo_to_csr_c` builds the CSR from the links (weights.py:31-39), the
library's `export_csr()` must equal it bit for bit -- once for the links in CDO's (dst, src) order (the builder's
sort-free path) and once for the same links shuffled (the bucketed-sort path, automatic thread count) -- before
it is handed to `oracle.apply_c` (at these sizes the kernels used to be checked against
the library's own matrix).
"""
import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import OperatorGroup, SparseOperator, _lib, gridgen, to_device
from smmregrid_amd.device import DeviceArray
from tests.helpers import assert_same

pytestmark = pytest.mark.gpu


def _operator(w, device=0):
    op = SparseOperator(w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values,
                        w["dst_address"].values, w["remap_matrix"].values, device=device)
    op.set_epilogue(w["dst_grid_imask"].values, w["dst_grid_frac"].values)
    return op


def _links(w, lev=None):
    """(src_address, dst_address, weight column 0) of a weights file, one level's `[:link_length]` of a 3-D one."""
    src, dst, rm = w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values
    if lev is not None:
        n = int(w["link_length"].values[lev])
        src, dst, rm = src[lev, :n], dst[lev, :n], rm[lev, :n]
    return src, dst, np.ascontiguousarray(rm[:, 0] if rm.ndim == 2 else rm)


def _assert_csr_bits(got, ref, what):
    assert np.array_equal(got[0], ref[0]), f"{what}: rowptr differs from the oracle's"
    assert np.array_equal(got[1], ref[1]), f"{what}: columns differ from the oracle's"
    assert np.array_equal(got[2].view(np.uint64), ref[2].view(np.uint64)), f"{what}: weights differ from the oracle's"


def _oracle_csr_checked(op, n_src, n_dst, src, dst, wgt, what, seed=0):
    """The oracle's own CSR of these links; `op` (built from the links as given: CDO order, the sort-free path)
    and a second operator built from the same links SHUFFLED (bucketed stable sort, automatic thread count)
    must both export exactly it.  The links of these files hold no duplicate coordinate (asserted), so the
    canonical CSR does not depend on the link order and one oracle build serves both."""
    ordered = np.all((dst[1:] > dst[:-1]) | ((dst[1:] == dst[:-1]) & (src[1:] >= src[:-1])))
    assert ordered, f"{what}: links are expected in (dst, src) order as cdo writes them"
    ref = oracle.coo_to_csr_c(n_src, n_dst, src, dst, wgt)
    assert ref[1].size == src.size, f"{what}: duplicate coordinates"
    _assert_csr_bits(op.export_csr(), ref, what + " (links in CDO order)")
    perm = np.random.default_rng(20260723 + seed).permutation(src.size)
    _lib.call("smm_set_host_threads", 0, None)                 # automatic thread count
    shuffled = SparseOperator(n_src, n_dst, src[perm], dst[perm], wgt[perm], device=0)
    try:
        _assert_csr_bits(shuffled.export_csr(), ref, what + " (links shuffled)")
    finally:
        shuffled.close()
    return ref


def _device_field(n_batch, n_src, dtype, seed, poke_rows):
    """(B, S) pseudo-normal field made on the device; `poke_rows` get NaN / +-inf from the host.
    Returns the DeviceArray and {row: host copy} for the poked rows."""
    x = DeviceArray((n_batch, n_src), dtype).fill_random(seed=seed, mean=250.0, sigma=30.0)
    rng = np.random.default_rng(seed)
    host = {}
    for r in poke_rows:
        row = x.rows(r, r + 1).to_host()[0]
        bad = rng.integers(0, n_src, size=max(4, n_src // 997))
        row[bad[0::3]] = np.nan
        row[bad[1::3]] = np.inf
        row[bad[2::3]] = -np.inf
        x.rows(r, r + 1).copy_from_host(row[None])
        host[r] = row
    return x, host


def _rows(x, rows):
    return np.stack([x.rows(r, r + 1).to_host()[0] for r in rows])


def _assert_device_equal(ya, yb, chunk=32):
    """Bit equality of two (B, D) device arrays (NaN-for-NaN), copied back chunk by chunk."""
    assert ya.shape == yb.shape
    for r0 in range(0, ya.shape[0], chunk):
        r1 = min(ya.shape[0], r0 + chunk)
        a = ya.rows(r0, r1).to_host().view(np.uint64 if ya.dtype == np.float64 else np.uint32)
        b = yb.rows(r0, r1).to_host().view(a.dtype)
        assert np.array_equal(a, b), f"tile and SELL kernels differ in batch rows {r0}..{r1 - 1}"


def _check_2d(op, csr, x, host_rows, sample, masked, amin, imask, frac):
    """default (tile) kernel vs oracle (on the oracle's own CSR `csr`) on `sample` rows, vs SELL kernel on every row."""
    y = op.apply(x, masked=masked, remap_area_min=amin)
    ys = op.apply(x, masked=masked, remap_area_min=amin, flags=_lib.APPLY_KERNEL_SELL)
    _assert_device_equal(y, ys)
    xs = np.stack([host_rows[r] if r in host_rows else x.rows(r, r + 1).to_host()[0] for r in sample])
    ref = oracle.apply_c(csr, xs, masked, imask, frac, amin, threads=8)
    assert_same(_rows(y, sample), ref, exact=True)
    return y


# ----------------------------------------------------------------- config 2: r1440x721 -> r360x180 bilinear, f64

def test_config2_production_walk(hip):
    w = gridgen.bilinear_weights("r1440x721", "r360x180")
    op = _operator(w)
    csr = _oracle_csr_checked(op, op.n_src, op.n_dst, *_links(w), "config 2", seed=2)
    B = 522                                                    # 131 batch tiles of 4 rows, last one ragged
    info = op.launch_info(B, np.float64)
    assert info["kernel"] == "tile" and info["j_per_block"] == 4 and info["rows_per_block"] == 256
    assert info["n_blocks"] == ((64800 + 255) // 256) * ((B + 3) // 4)
    x, host = _device_field(B, op.n_src, np.float64, 11, poke_rows=[1, 257, B - 1])
    _check_2d(op, csr, x, host, [0, 1, 2, 3, 4, 257, 300, B - 2, B - 1], False, 0.5, None, w["dst_grid_frac"].values)


# ----------------------------------------------------------------- config 5: r1440x721 -> r720x360 conservative, f64

def test_config5_production_walk(hip):
    w = gridgen.conservative_weights("r1440x721", "r720x360")
    op = _operator(w)
    csr = _oracle_csr_checked(op, op.n_src, op.n_dst, *_links(w), "config 5", seed=5)
    B = 520                                                    # 9 batch tiles of 64 rows, last one 8
    info = op.launch_info(B, np.float64)
    assert info["kernel"] == "tile" and info["j_per_block"] == 64
    x, host = _device_field(B, op.n_src, np.float64, 12, poke_rows=[63, 64, B - 1])
    _check_2d(op, csr, x, host, [0, 63, 64, 65, 127, 128, 300, 511, 512, B - 1], False, 0.5, None,
              w["dst_grid_frac"].values)


# ----------------------------------------------------------------- config 4: n1280 -> HEALPix 1024 bilinear, f32 in

def test_config4_geometry_f32(hip):
    """Gaussian n1280 (5120x2560) -> HEALPix nside 1024, 50 M links, f32 in / f64 out: the paths
    only this geometry takes -- an operator larger than L2 (walks of 128 batch rows), LDS-DMA staging
    (or multi-row register steps) on its small tiles, the tightened tile budget with the widest blocks
    demoted to direct gathers."""
    w = gridgen.generate_weights("n1280", "hp1024", method="bil")
    assert (w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w.sizes["num_links"]) == \
        (13107200, 12582912, 50331648)
    op = _operator(w)
    # all 50 M links: the oracle's own CSR against the sort-free build and against the bucketed sort of the shuffled links
    csr = _oracle_csr_checked(op, op.n_src, op.n_dst, *_links(w), "config 4", seed=4)
    plan = op.plan_info()
    assert plan["tile_plan"] and plan["tile_preferred"] and plan["rows_per_block"] == 256
    # tighten_tile_plan shrank the tile below half of the 64-KiB chunk budget (polar caps -> direct)
    assert 0 < plan["lds_bytes"] <= 32768
    B = 130                                                    # one full walk of 128 rows + a ragged one
    info = op.launch_info(B, np.float32)
    # small tiles of a 16-B aligned field: staged by LDS-DMA into a ring of two slots, single-row steps
    assert info["kernel"] == "tile-dma" and info["big_operator"] and info["j_per_block"] == 128
    assert info["rows_per_step"] == 1 and info["lds_bytes"] <= 16384
    x, host = _device_field(B, op.n_src, np.float32, 13, poke_rows=[1, 129])
    y = _check_2d(op, csr, x, host, [0, 1, 127, 128, 129], False, 0.0, None, None)
    assert y.dtype == np.float64                               # result_type(f32, f64), regrid.py:550
    # register staging with multi-row steps (R = 2 or 4) serves fields that are not 16-B aligned and is
    # what the tuning knob tile_staging = 1 forces: same bits
    with _lib.tuning(tile_staging=_lib.STAGING_REGISTERS):
        info8 = op.launch_info(B, np.float32)
        assert info8["kernel"] == "tile" and info8["rows_per_step"] > 1
        _assert_device_equal(op.apply(x), y)
    # a short batch takes the same kernels with a clipped walk
    x8 = x.rows(0, 8)
    y8 = op.apply(x8)
    _assert_device_equal(y8, y.rows(0, 8))


# ----------------------------------------------------------------- config 3: 75 masked ocean levels, grouped launch

@pytest.fixture(scope="module")
def config3():
    nx, ny, n_lev = 1442, 1021, 75
    src = gridgen.regular_grid(nx, ny)
    masks = gridgen.synthetic_ocean_masks(nx, ny, n_lev)
    levels = np.arange(n_lev, dtype=np.float64)
    w3 = gridgen.ConservativeLevels(src, "r360x180").stack(masks, levels)
    ll = w3["link_length"].values
    ops, csrs = [], []
    imask = np.empty((n_lev, 64800), np.int32)
    for i in range(n_lev):
        op = SparseOperator(src.size, 64800, w3["src_address"].values[i, :ll[i]],
                            w3["dst_address"].values[i, :ll[i]], w3["remap_matrix"].values[i, :ll[i], 0], device=0)
        imask[i] = op.mask_apply(masks[i])
        # every level's CSR is the oracle's own, built from that level's links (weights.py:7-23); levels 0, 37, 74
        # are also rebuilt from shuffled links
        lk = _links(w3, i)
        if i in (0, 37, 74):
            csrs.append(_oracle_csr_checked(op, src.size, 64800, *lk, f"config 3 level {i}", seed=300 + i))
        else:
            csrs.append(oracle.coo_to_csr_c(src.size, 64800, *lk))
            _assert_csr_bits(op.export_csr(), csrs[i], f"config 3 level {i}")
        assert np.array_equal(imask[i], oracle.mask_apply_c(csrs[i], masks[i]))    # weights.py:47-52
        op.set_epilogue(imask[i], w3["dst_grid_frac"].values[i])
        ops.append(op)
    grp = OperatorGroup(ops)
    return {"src": src, "masks": masks, "w3": w3, "ops": ops, "csrs": csrs, "imask": imask,
            "frac": w3["dst_grid_frac"].values, "grp": grp, "n_lev": n_lev,
            "masked_levels": (~(imask == 1).all(axis=1)).astype(np.uint8)}


def _slabs(c, n, seed):
    rng = np.random.default_rng(seed)
    S = c["src"].size
    slabs = (10.0 + 5.0 * rng.standard_normal((n, c["n_lev"], S), dtype=np.float32)).astype(np.float64)
    slabs[:, c["masks"] == 0] = np.nan                         # land per level
    return slabs


def test_config3_all_levels_production_walk(hip, config3):
    """All 75 levels in one grouped launch, 66 time steps (one full walk of 64 + a ragged one).  Time
    steps cycle irregularly through 3 distinct slabs, every (t, level) row is compared with the
    oracle's result for its slab."""
    c = config3
    grp, L, S, D = c["grp"], c["n_lev"], c["src"].size, 64800
    assert max(o.max_row_nnz for o in c["ops"]) > 32 and grp.plan_info()["tile_plan"]
    T = 66
    info = grp.launch_info(T, L, 1)
    assert info["kernel"] == "tile" and info["j_per_block"] == 64 and info["rows_per_block"] == 64
    slabs = _slabs(c, 3, 21)
    order = np.random.default_rng(5).integers(0, 3, size=T)
    order[:3] = [0, 1, 2]
    dslab = [to_device(s.reshape(1, L, 1, S)) for s in slabs]
    x = DeviceArray((T, L, 1, S), np.float64)
    import ctypes
    for t in range(T):
        _lib.call("smm_memcpy_d2d", ctypes.c_void_p(x.rows(t, t + 1).ptr), ctypes.c_void_p(dslab[order[t]].ptr),
                  slabs[0].nbytes, None)
    level_index = np.arange(L, dtype=np.int32)
    y = grp.apply(x, level_index, c["masked_levels"], masked=True, remap_area_min=0.5, transpose=True)
    ys = grp.apply(x, level_index, c["masked_levels"], masked=True, remap_area_min=0.5, transpose=True,
                   flags=_lib.APPLY_KERNEL_SELL)
    yh, ysh = y.to_host(), ys.to_host()                        # (T, 1, L, D)
    assert np.array_equal(yh.view(np.uint64), ysh.view(np.uint64))
    refs = [oracle.apply_levels(c["csrs"], s[None], 1, level_index, c["masked_levels"].astype(bool),
                                c["imask"], c["frac"], 0.5, transpose=True) for s in slabs]   # (1, L, D)
    for t in range(T):
        assert_same(yh[t, 0], refs[order[t]][0], exact=True)


@pytest.mark.parametrize("transpose", [True, False])
def test_config3_level_subset(hip, config3, transpose):
    """levels_test.py:10-27 at full size: data carrying levels [14, 15, 17] (and [15] alone) picks the
    matching operators out of the 75; T = 2."""
    c = config3
    grp, S = c["grp"], c["src"].size
    slabs = _slabs(c, 2, 22)                                   # (T=2, L, S)
    for sel in ([14, 15, 17], [15], [74, 0]):
        x = np.ascontiguousarray(slabs[:, sel])[:, :, None, :]     # (2, len(sel), 1, S)
        lev = np.asarray(sel, np.int32)
        y = grp.apply(to_device(x), lev, c["masked_levels"], masked=True, remap_area_min=0.5,
                      transpose=transpose).to_host()
        ref = oracle.apply_levels(c["csrs"], x, 1, lev, c["masked_levels"].astype(bool), c["imask"],
                                  c["frac"], 0.5, transpose=transpose)
        assert_same(y, ref, exact=True)
