"""Robustness of the handle / pipeline code around the kernels: level-configuration cache under
concurrent threads, graph capture of the grouped apply, group membership rules, and the error
path of the host pipelines."""
import ctypes
import os
import threading

import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import OperatorGroup, SparseOperator, _lib, pinned_empty, to_device
from smmregrid_amd.device import DeviceArray, Stream
from tests.helpers import assert_same, field, random_links

pytestmark = pytest.mark.gpu


def _group(rng, S=700, D=190, L=6):
    ops, csrs = [], []
    imask = (rng.random((L, D)) > 0.3).astype(np.int32)
    frac = rng.random((L, D))
    for l in range(L):
        src, dst, w = random_links(rng, S, D, 500 + 60 * l)
        op = SparseOperator(S, D, src, dst, w, device=0)
        op.set_epilogue(imask[l], frac[l])
        ops.append(op)
        csrs.append(op.export_csr())
    return OperatorGroup(ops), ops, csrs, imask, frac


def test_level_config_cache_under_thread_hammer(hip, rng):
    """More than 64 distinct level subsets cycled by several threads at once (round 1 flushed the
    cache at 64 entries with a device-wide synchronise + free while other threads could still hold
    an entry): every result stays bit-identical to the oracle."""
    S, D, L = 700, 190, 6
    grp, ops, csrs, imask, frac = _group(rng, S, D, L)
    n_threads, n_iter = 6, 40
    # 6 x 40 = 240 calls over ~150 distinct (level_index, masked_levels) configurations
    configs = []
    crng = np.random.default_rng(3)
    for _ in range(150):
        n_lev = int(crng.integers(1, 5))
        lev = crng.integers(0, L, size=n_lev).astype(np.int32)
        ml = crng.integers(0, 2, size=L).astype(np.uint8) if crng.random() < 0.7 else None
        configs.append((lev, ml))
    xs = {n: field(rng, 3 * n, S, nan_frac=0.02).reshape(3, n, 1, S) for n in range(1, 5)}
    refs = {}
    for i, (lev, ml) in enumerate(configs):
        mlb = np.ones(L, bool) if ml is None else ml.astype(bool)
        refs[i] = oracle.apply_levels(csrs, xs[lev.size], 1, lev, mlb, imask, frac, 0.4, True)
    errors = []

    def work(tid):
        try:
            s = Stream()
            dxs = {n: to_device(x) for n, x in xs.items()}
            trng = np.random.default_rng(100 + tid)
            for _ in range(n_iter):
                i = int(trng.integers(0, len(configs)))
                lev, ml = configs[i]
                dy = grp.apply(dxs[lev.size], lev, ml, masked=True, remap_area_min=0.4, transpose=True, stream=s)
                s.synchronize()
                assert_same(dy.to_host(), refs[i], exact=True)
        except Exception as exc:  # pragma: no cover
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:2]


def test_level_config_key_is_unambiguous(hip, rng):
    """n levels + a masked_levels vector vs n+1 levels without one must not share a cache entry
    (the round-1 key concatenated them without the level count)."""
    S, D, L = 700, 190, 4
    grp, ops, csrs, imask, frac = _group(rng, S, D, L)
    x3 = field(rng, 3, S).reshape(1, 3, 1, S)
    x4 = field(rng, 4, S).reshape(1, 4, 1, S)
    # bytes of level_index [0,0,0] + flag 1 + masked [0,0,0,0]  ==  level_index [0,0,0,X] ... for a
    # little-endian X whose bytes are (1,0,0,0): X = 1 -> second call uses 4 levels [0,0,0,1], no mask
    ya = grp.apply(to_device(x3), [0, 0, 0], np.zeros(L, np.uint8), masked=True, transpose=True).to_host()
    yb = grp.apply(to_device(x4), [0, 0, 0, 1], None, masked=True, transpose=True).to_host()
    ra = oracle.apply_levels(csrs, x3, 1, [0, 0, 0], np.zeros(L, bool), imask, frac, 0.0, True)
    rb = oracle.apply_levels(csrs, x4, 1, [0, 0, 0, 1], np.ones(L, bool), imask, frac, 0.0, True)
    assert_same(ya, ra, exact=True)
    assert_same(yb, rb, exact=True)


def test_group_apply_is_graph_capturable_after_prepare(hip, rng):
    """smm_group_prepare uploads the level configuration; the apply then allocates nothing and never
    blocks, so it can be captured into a hipGraph."""
    hiprt = ctypes.CDLL("libamdhip64.so.7")
    S, D, L = 700, 190, 5
    grp, ops, csrs, imask, frac = _group(rng, S, D, L)
    lev = np.array([4, 1, 2], np.int32)
    ml = np.array([1, 0, 1, 1, 1], np.uint8)
    x = field(rng, 6, S, nan_frac=0.02).reshape(2, 3, 1, S)
    dx = to_device(x)
    dy = DeviceArray((2, 1, 3, D), np.float64).fill_bytes(0)
    s = Stream()
    grp.prepare(lev, ml)
    graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
    assert hiprt.hipStreamBeginCapture(s.handle, 0) == 0
    grp.apply(dx, lev, ml, y=dy, masked=True, remap_area_min=0.4, transpose=True, stream=s)
    assert hiprt.hipStreamEndCapture(s.handle, ctypes.byref(graph)) == 0
    assert hiprt.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, ctypes.c_size_t(0)) == 0
    for _ in range(3):
        assert hiprt.hipGraphLaunch(gexec, s.handle) == 0
    s.synchronize()
    ref = oracle.apply_levels(csrs, x, 1, lev, ml.astype(bool), imask, frac, 0.4, True)
    assert_same(dy.to_host(), ref, exact=True)
    hiprt.hipGraphExecDestroy(gexec)
    hiprt.hipGraphDestroy(graph)


def test_group_members_are_frozen(hip, rng):
    """A group's level table holds its members' imask / frac device pointers: changing a member's
    epilogue or destroying it while the group lives is refused; afterwards both work again."""
    S, D, L = 700, 190, 3
    grp, ops, csrs, imask, frac = _group(rng, S, D, L)
    with pytest.raises(_lib.SmmError) as e:
        ops[1].set_epilogue(imask[0], frac[0])
    assert e.value.code == _lib.SMM_ERR_INVALID
    assert _lib.load().smm_operator_destroy(ops[1].handle) == _lib.SMM_ERR_INVALID
    x = field(rng, 3, S).reshape(1, 3, 1, S)
    y = grp.apply(to_device(x), [0, 1, 2], masked=True, remap_area_min=0.3).to_host()
    assert_same(y, oracle.apply_levels(csrs, x, 1, [0, 1, 2], np.ones(L, bool), imask, frac, 0.3, True), exact=True)
    grp.close()
    ops[1].set_epilogue(imask[0], frac[0])          # allowed again
    y1 = ops[1].apply(to_device(x[0, 1]), masked=True, remap_area_min=0.3).to_host()
    assert_same(y1, oracle.apply_c(csrs[1], x[0, 1], True, imask[0], frac[0], 0.3), exact=True)


@pytest.mark.parametrize("pinned", [False, True])
def test_host_pipeline_error_drains_in_flight_copies(hip, rng, pinned):
    """A failure at chunk c must not return while chunk c-1's asynchronous copy into the caller's Y
    is still in flight: after the error every row of the chunks before c is complete."""
    S, D, B, chunk = 20000, 6000, 48, 8
    src, dst, w = random_links(rng, S, D, 40000)
    op = SparseOperator(S, D, src, dst, w, device=0)
    x = field(rng, B, S)
    ref = oracle.apply_c(op.export_csr(), x)
    xin = x
    out = np.full((B, D), -1.0)
    if pinned:
        xin = pinned_empty(x.shape, np.float64)
        xin[...] = x
        out = pinned_empty((B, D), np.float64)
        out[...] = -1.0
    fail_at = 3
    _lib.call("smm_debug_fail_at_chunk", fail_at)      # explicit test hook, not an environment variable
    try:
        with pytest.raises(_lib.SmmError) as e:
            op.apply_host(xin, out=out, chunk_rows=chunk)
        assert "injected failure" in str(e.value)
    finally:
        _lib.call("smm_debug_fail_at_chunk", -1)
    # direct DMA (pinned): chunks 0..c-1 were enqueued and must all have landed before the return;
    # staged copies are delivered by drain(), which had handled chunks 0..c-2 when chunk c failed
    done = fail_at * chunk if pinned else (fail_at - 1) * chunk
    got = np.array(out)
    assert_same(got[:done], ref[:done], exact=True)
    assert (got[fail_at * chunk:] == -1.0).all()                 # nothing past the failing chunk was touched
    # the pipeline is usable again afterwards
    y = op.apply_host(xin, chunk_rows=chunk)
    assert_same(y, ref, exact=True)


def test_group_pipeline_error_path_and_whole_call_validation(hip, rng):
    """The level-group host pipeline: an injected failure at chunk c returns an error (nothing hangs, the
    pipeline is usable afterwards); and a call whose LAST level lacks dst_frac is refused before any level
    has written to Y -- by the device-resident batch-fastest entry and by the host pipeline's pack branch."""
    S, D, L = 3000, 260, 3
    ops, csrs = [], []
    imask = (rng.random((L, D)) > 0.3).astype(np.int32)
    frac = rng.random((L, D))
    for i in range(L):
        src, dst, w = random_links(rng, S, D, 400 + 100 * i)          # few used cells: the pack branch qualifies
        op = SparseOperator(S, D, src, dst, w, device=0)
        op.set_epilogue(imask[i], frac[i] if i < L - 1 else None)      # the last level has no dst_frac
        ops.append(op)
        csrs.append(op.export_csr())
    grp = OperatorGroup(ops)
    x = field(rng, 40 * L, S).reshape(40, L, 1, S)
    lev = np.arange(L, dtype=np.int32)
    # remap_area_min > 0 needs every selected level's dst_frac: refused as a whole
    xsb = to_device(np.ascontiguousarray(x.reshape(40, L, S).transpose(1, 2, 0)), layout="sb")
    y = DeviceArray((40, L, D), np.float64).fill_bytes(0)
    with pytest.raises(_lib.SmmError) as e:
        grp.apply_sb(xsb, lev, y=y, remap_area_min=0.4)
    assert e.value.code == _lib.SMM_ERR_INVALID and "dst_frac" in str(e.value)
    assert (y.to_host() == 0.0).all()                                  # no level was launched
    with pytest.raises(_lib.SmmError):
        grp.apply_host(x, lev, remap_area_min=0.4)
    # without the threshold the same calls work, and an injected chunk failure surfaces as an error
    ref = oracle.apply_levels(csrs, x, 1, lev, np.ones(L, bool), imask, None, 0.0, True)
    assert_same(grp.apply_host(x, lev, masked=True), ref, exact=True)
    _lib.call("smm_debug_fail_at_chunk", 0)
    try:
        with pytest.raises(_lib.SmmError) as e:
            grp.apply_host(x, lev, masked=True)
        assert "injected failure" in str(e.value)
    finally:
        _lib.call("smm_debug_fail_at_chunk", -1)
    assert_same(grp.apply_host(x, lev, masked=True), ref, exact=True)


def test_pitched_copies_and_their_errors(hip, rng):
    """smm_memcpy2d_h2d / _d2h: rows of `width` bytes on different pitches; a pitch below the width is refused."""
    import ctypes
    rows, cols, pitch = 7, 13, 24
    host = rng.standard_normal((rows, cols))
    dev = DeviceArray((rows, pitch), np.float64).fill_bytes(0)
    _lib.call("smm_memcpy2d_h2d", ctypes.c_void_p(dev.ptr), pitch * 8, host.ctypes.data_as(ctypes.c_void_p), cols * 8,
              cols * 8, rows, None)
    full = dev.to_host()
    assert np.array_equal(full[:, :cols], host) and (full[:, cols:] == 0.0).all()
    back = np.zeros((rows, cols))
    _lib.call("smm_memcpy2d_d2h", back.ctypes.data_as(ctypes.c_void_p), cols * 8, ctypes.c_void_p(dev.ptr), pitch * 8,
              cols * 8, rows, None)
    assert np.array_equal(back, host)
    col = np.zeros(rows)                                               # one column: width 8 B
    _lib.call("smm_memcpy2d_d2h", col.ctypes.data_as(ctypes.c_void_p), 8, ctypes.c_void_p(dev.ptr + 3 * 8), pitch * 8,
              8, rows, None)
    assert np.array_equal(col, host[:, 3])
    with pytest.raises(_lib.SmmError) as e:
        _lib.call("smm_memcpy2d_h2d", ctypes.c_void_p(dev.ptr), 8, host.ctypes.data_as(ctypes.c_void_p), cols * 8,
                  cols * 8, rows, None)
    assert e.value.code == _lib.SMM_ERR_INVALID
