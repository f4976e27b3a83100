"""Round 6: the ABI refuses apply-flag bits it does not define; nothing a staging task throws crosses the C ABI; the
host pipelines report where their time went; the pack's store kind and the staging thread count do not change a bit."""
import ctypes

import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import OperatorGroup, SparseOperator, _lib, gridgen, pinned_empty, to_device
from smmregrid_amd.device import DeviceArray
from tests.helpers import assert_same, field, random_links

pytestmark = pytest.mark.gpu


def _bilinear(src="r288x144", dst="r72x36"):
    w = gridgen.bilinear_weights(src, dst)
    op = SparseOperator(w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values,
                        w["dst_address"].values, w["remap_matrix"].values, device=0)
    return w, op


def test_unknown_apply_flag_bits_are_refused(hip, rng):
    """ABI v4 encoded kernel variants in bits 16..27 of `flags`; v5 moved them to smm_debug_set_tuning and v6 refuses
    what is left over, in every entry that takes flags, so that a stale caller fails loudly instead of silently
    running the default kernel."""
    w, op = _bilinear()
    x = field(rng, 40, op.n_src)
    xd = to_device(x)
    for bad in (1 << 16, 15 << 16, 3 << 20, 1 << 5, 1 << 31):
        for call in (lambda: op.apply(xd, flags=bad),
                     lambda: op.apply(xd, flags=bad | _lib.APPLY_KERNEL_TILE),
                     lambda: op.apply_sb(to_device(np.ascontiguousarray(x.T)), flags=bad),
                     lambda: op.apply_host(x, flags=bad),
                     lambda: op.launch_info(40, flags=bad)):
            with pytest.raises(_lib.SmmError) as e:
                call()
            assert e.value.code == _lib.SMM_ERR_INVALID and "unknown apply flag bits" in str(e.value)
    grp = OperatorGroup([op])
    with pytest.raises(_lib.SmmError) as e:
        grp.apply(to_device(x.reshape(40, 1, 1, -1)), [0], flags=1 << 16)
    assert e.value.code == _lib.SMM_ERR_INVALID
    with pytest.raises(_lib.SmmError):
        grp.apply_host(x.reshape(40, 1, 1, -1), [0], flags=1 << 20)
    # every defined bit is still accepted
    ok = op.apply(xd, flags=_lib.APPLY_KERNEL_SELL | _lib.APPLY_NO_FILL).to_host()
    assert_same(ok, oracle.apply_c(op.export_csr(), x), exact=True)
    grp.close()


@pytest.mark.parametrize("mode", ["packed", "whole_rows"])
def test_staging_task_failure_returns_a_status_and_drains(hip, rng, mode):
    """A std::bad_alloc thrown inside a staging task (pack / copy-in on the pool's threads) comes back from
    smm_apply_host as SMM_ERR_ALLOC -- no std::terminate inside extern "C" -- after the copies in flight into the
    caller's Y have landed; with thread start refused the calling thread stages alone and the bits are the same; the
    pipeline is usable afterwards."""
    w, op = _bilinear("r1440x720", "r360x180")       # U = S / 4: the packing variant applies; 8.3 MB per row
    S, D = op.n_src, op.n_dst
    B, chunk = 192, 32
    x = field(rng, B, S)
    ref = oracle.apply_c(op.export_csr(), x)
    fl = 0 if mode == "packed" else _lib.APPLY_HOST_NO_PACK
    assert_same(op.apply_host(x, flags=fl, chunk_rows=chunk), ref, exact=True)
    try:
        _lib.call("smm_debug_staging_faults", 1, -1)                 # no worker thread can be started
        assert_same(op.apply_host(x, flags=fl, chunk_rows=chunk), ref, exact=True)
        codes = []
        for no_threads in (0, 1):
            for at in (0, 3, 40):
                out = np.full((B, D), -1.0)
                _lib.call("smm_debug_staging_faults", no_threads, at)
                try:
                    op.apply_host(x, out=out, flags=fl, chunk_rows=chunk)
                    codes.append(0)                                    # the counter ran past every task of this call
                    assert_same(out, ref, exact=True)
                except _lib.SmmError as e:
                    codes.append(e.code)
                    assert e.code == _lib.SMM_ERR_ALLOC and "staging task" in str(e)
                    rows = out[:, 0] != -1.0                           # delivered chunks are complete and right
                    assert_same(out[rows], ref[rows], exact=True)
        assert _lib.SMM_ERR_ALLOC in codes
    finally:
        _lib.call("smm_debug_staging_faults", 0, -1)
    assert_same(op.apply_host(x, flags=fl, chunk_rows=chunk), ref, exact=True)


def test_host_pipeline_stats_and_store_kinds(hip, rng):
    """smm_debug_host_stats: one call = one `calls`, its chunks, wall time >= every host-side stage, event times of H2D /
    kernel / D2H present; pinned input and output have no copy stages.  The pack gives the same bits with non-temporal
    and plain stores, 1 and 8 staging threads."""
    w, op = _bilinear("r1440x720", "r360x180")
    S, D = op.n_src, op.n_dst
    B = 256
    x = field(rng, B, S)
    ref = oracle.apply_c(op.export_csr(), x)
    op.apply_host(x)                                                   # warm-up: staging buffers
    _lib.host_stats(reset=True)
    y = op.apply_host(x, out=np.empty((B, D)), chunk_rows=64)        # an ordinary (pageable) result array: a copy-out stage
    st = _lib.host_stats(reset=True)
    assert_same(y, ref, exact=True)
    assert st["calls"] == 1 and st["chunks"] == 4 and st["threads"] >= 1
    assert st["stage_in_ms"] > 0 and st["copy_out_ms"] > 0 and st["h2d_ms"] > 0 and st["kernel_ms"] > 0 and st["d2h_ms"] > 0
    assert st["total_ms"] >= max(st["stage_in_ms"], st["copy_out_ms"], st["wait_ms"]) * 0.999
    assert _lib.host_stats()["calls"] == 0                              # reset
    xp, yp = pinned_empty((B, S), np.float64), pinned_empty((B, D), np.float64)
    xp[...] = x
    op.apply_host(xp, out=yp, flags=_lib.APPLY_HOST_NO_PACK, chunk_rows=64)
    st = _lib.host_stats(reset=True)
    assert_same(np.array(yp), ref, exact=True)
    assert st["stage_in_ms"] == 0 and st["copy_out_ms"] == 0 and st["h2d_ms"] > 0 and st["chunks"] == 4
    prev = ctypes.c_int(0)
    for threads in (1, 8):
        _lib.call("smm_set_host_threads", threads, ctypes.byref(prev))
        try:
            for stores in (0, 1):
                with _lib.tuning(host_pack_stores=stores):
                    assert_same(op.apply_host(x, chunk_rows=96), ref, exact=True)      # 96 + 96 + 64 rows
                    assert_same(op.apply_host(x[:37]), ref[:37], exact=True)           # a short, odd chunk
        finally:
            _lib.call("smm_set_host_threads", prev.value, None)
    assert _lib.host_stats()["threads"] in (1.0, 8.0)


def test_concurrent_host_pipelines_share_the_staging_pool(hip, rng):
    """The reference's dask scheduler calls apply_weights from several threads (regrid.py:29-30): four threads run
    smm_apply_host on two operators at once (calls on one operator take turns, the staging jobs of different
    operators take turns on the one pool) while a fifth thread builds operators (the builders' own worker threads).
    Every result is bit-equal to the oracle."""
    import threading
    ops = [_bilinear("r1440x720", "r360x180")[1], _bilinear("r720x360", "r180x90")[1]]
    xs = [field(rng, 160, op.n_src) for op in ops]
    refs = [oracle.apply_c(op.export_csr(), x) for op, x in zip(ops, xs)]
    errors = []

    def apply_loop(k, reps):
        try:
            for r in range(reps):
                y = ops[k].apply_host(xs[k], flags=_lib.APPLY_HOST_NO_PACK if (r + k) % 2 else 0, chunk_rows=48)
                if not np.array_equal(y, refs[k], equal_nan=True):
                    errors.append(f"operator {k}, call {r}: result differs from the oracle")
        except Exception as exc:          # noqa: BLE001 -- reported through `errors`
            errors.append(repr(exc))

    def create_loop():
        try:
            for _ in range(3):
                w, op = _bilinear("r720x360", "r90x45")
                x = field(rng, 3, op.n_src)
                if not np.array_equal(op.apply(to_device(x)).to_host(), oracle.apply_c(op.export_csr(), x), equal_nan=True):
                    errors.append("operator created under load differs")
                op.close()
        except Exception as exc:          # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=apply_loop, args=(k % 2, 4)) for k in range(4)] + [threading.Thread(target=create_loop)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not any(t.is_alive() for t in threads), "a host pipeline call did not return"
    assert errors == [], errors[:3]


@pytest.mark.parametrize("transpose", [True, False])
@pytest.mark.parametrize("n_inner,dtype", [(1, np.float64), (2, np.float32)])
def test_level_major_packed_chunks_of_the_group_pipeline(hip, rng, transpose, n_inner, dtype):
    """smm_group_apply_host, round 6: when 32 batch entries of ALL selected levels do not fit the staging budget (config 3:
    10 GB), the packed chunks become level-major -- a few consecutive data levels x a block of the outer axis, Y copied
    back as the level range of those rows.  SMM_TUNE_HOST_CHUNK_KB forces that form at test sizes: several levels per
    chunk, one level per chunk, several blocks of the outer axis with a short last one; pageable and pinned Y; both Y
    layouts; a level subset in another order.  Every result bit-equal to the oracle and to the whole-row pipeline."""
    from smmregrid_amd.weights import compute_weights_matrix3d
    nx, ny, n_lev = 96, 48, 6
    src = gridgen.regular_grid(nx, ny)
    masks = gridgen.synthetic_ocean_masks(nx, ny, n_lev, top=0.45, bottom=0.06)
    w3 = gridgen.ConservativeLevels(src, "r24x12").stack(masks, np.arange(n_lev, dtype=np.float64))
    ops = compute_weights_matrix3d(w3, "lev", device=0)
    S, D = ops[0].n_src, ops[0].n_dst
    imask = np.stack([op.mask_apply(masks[i]) for i, op in enumerate(ops)])
    frac = w3["dst_grid_frac"].values
    for i, op in enumerate(ops):
        op.set_epilogue(imask[i], frac[i])
    assert sum(op.n_used_src for op in ops) * 2 <= n_lev * S          # the packing variant applies
    grp = OperatorGroup(ops)
    csrs = [op.export_csr() for op in ops]
    masked_levels = (~(imask == 1).all(axis=1)).astype(np.uint8)
    for level_index in (np.arange(n_lev, dtype=np.int32), np.array([4, 1, 2], dtype=np.int32)):
        nl = level_index.size
        n_outer = 150
        x = (10.0 + 5.0 * rng.standard_normal((n_outer, nl, n_inner, S))).astype(dtype)
        for k, l in enumerate(level_index):
            x[:, k][:, :, masks[l] == 0] = np.nan
        ref = oracle.apply_levels(csrs, x, 1, level_index, masked_levels.astype(bool), imask, frac, 0.5, transpose)
        whole = grp.apply_host(x, level_index, masked_levels, masked=True, remap_area_min=0.5, transpose=transpose,
                               flags=_lib.APPLY_HOST_NO_PACK)
        assert_same(whole, ref, exact=True)
        for kb in (0, 4096, 640, 160):              # default rule; a few levels per chunk ... one level and a block of 38 rows per chunk
            with _lib.tuning(host_chunk_kb=kb):
                _lib.host_stats(reset=True)
                y = grp.apply_host(x, level_index, masked_levels, masked=True, remap_area_min=0.5, transpose=transpose)
                st = _lib.host_stats(reset=True)
            assert_same(y, ref, exact=True)
            if kb == 160:
                assert st["chunks"] >= 4 * nl             # level-major AND four blocks of the outer axis (38, 38, 38, 36 rows)
    # a short batch (10 entries: below the 32 a block of all levels needs, above the 8 level-major chunks take) is packed too
    level_index = np.arange(n_lev, dtype=np.int32)
    x = (10.0 + 5.0 * rng.standard_normal((10, n_lev, n_inner, S))).astype(dtype)
    for l in range(n_lev):
        x[:, l][:, :, masks[l] == 0] = np.nan
    ref = oracle.apply_levels(csrs, x, 1, level_index, masked_levels.astype(bool), imask, frac, 0.5, transpose)
    _lib.host_stats(reset=True)
    y = grp.apply_host(x, level_index, masked_levels, masked=True, remap_area_min=0.5, transpose=transpose)
    st = _lib.host_stats(reset=True)
    assert_same(y, ref, exact=True)
    assert st["h2d_ms"] > 0 and st["stage_in_ms"] > 0
    # pinned Y: the level range of a chunk goes back by a pitched D2H copy
    level_index = np.arange(n_lev, dtype=np.int32)
    x = (10.0 + 5.0 * rng.standard_normal((64, n_lev, n_inner, S))).astype(dtype)
    ref = oracle.apply_levels(csrs, x, 1, level_index, masked_levels.astype(bool), imask, frac, 0.5, transpose)
    shape = (64, n_inner, n_lev, D) if transpose else (n_lev, 64, n_inner, D)
    out = pinned_empty(shape, np.float64)
    lev, ml = grp._level_args(level_index, masked_levels, n_lev)
    with _lib.tuning(host_chunk_kb=320):
        _lib.call("smm_group_apply_host", grp.handle, x.ctypes.data_as(ctypes.c_void_p), 1 if dtype == np.float64 else 0,
                  out.ctypes.data_as(ctypes.c_void_p), 1, 64, n_lev, n_inner, int(transpose),
                  lev.ctypes.data_as(ctypes.c_void_p), ml.ctypes.data_as(ctypes.c_void_p), 0.5, _lib.APPLY_MASKED, 0)
    assert_same(np.array(out), ref, exact=True)
    grp.close()


def test_results_come_in_recycled_page_locked_buffers(hip, rng):
    """apply_host returns a fresh array per call (as the reference's regrid does).  The first large result is an ordinary array
    (hipHostMalloc never runs on the caller's path) while a page-locked block of its size is prepared in the background; later
    results take recycled page-locked blocks -- written by the DMA engine directly, no first-touch page faults, no copy-out
    stage.  Results a caller still holds are never touched by later calls; small results are ordinary arrays;
    SMM_RESULT_CACHE=0 switches it off."""
    import gc
    import os
    from smmregrid_amd.device import result_cache
    w, op = _bilinear("r1440x720", "r360x180")
    x = field(rng, 256, op.n_src)                                  # Y: 256 x 64 800 x 8 B = 133 MB
    ref = oracle.apply_c(op.export_csr(), x)
    result_cache.clear()
    gc.collect()
    live0, hits0 = result_cache.live, result_cache.hits
    y0 = op.apply_host(x)                                          # no block yet: an ordinary array, a block is being prepared
    assert y0.flags.owndata
    assert_same(y0, ref, exact=True)
    result_cache.wait()
    assert result_cache.cached >= y0.nbytes and result_cache.live == live0
    _lib.host_stats(reset=True)
    y1 = op.apply_host(x)
    st = _lib.host_stats(reset=True)
    assert_same(y1, ref, exact=True)
    assert st["copy_out_ms"] == 0 and result_cache.hits == hits0 + 1            # written by the DMA engine: no staging copy
    assert result_cache.live == live0 + y1.nbytes and not y1.flags.owndata and y1.flags.writeable
    y2 = op.apply_host(x * 2.0)                                   # y1 is still held: its block is not reused, y1 untouched
    assert_same(y1, ref, exact=True)
    assert_same(y2, oracle.apply_c(op.export_csr(), x * 2.0), exact=True)
    keep = y1[3].copy()
    del y1
    gc.collect()
    result_cache.wait()
    assert result_cache.cached >= y2.nbytes                        # the block went back to the cache ...
    hits = result_cache.hits
    y3 = op.apply_host(x)
    assert result_cache.hits == hits + 1 and not y3.flags.owndata                          # ... and serves the next result
    assert np.array_equal(y3[3], keep, equal_nan=True)
    assert_same(y3, ref, exact=True)
    assert op.apply_host(x[:4]).flags.owndata                     # 2 MB: an ordinary array
    os.environ["SMM_RESULT_CACHE"] = "0"
    try:
        assert op.apply_host(x).flags.owndata
    finally:
        del os.environ["SMM_RESULT_CACHE"]
    del y0, y2, y3
    gc.collect()
    result_cache.clear()
