"""smm_apply allocates nothing and never synchronises, so it can be captured into a hipGraph and
replayed (launch-bound loops over many small fields)."""
import ctypes

import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import SparseOperator, gridgen, to_device
from smmregrid_amd.device import DeviceArray, Stream
from tests.helpers import assert_same, field

pytestmark = pytest.mark.gpu


def test_apply_is_graph_capturable(hip, rng):
    hiprt = ctypes.CDLL("libamdhip64.so.7")          # by SONAME: the copy the library itself is bound to
    w = gridgen.bilinear_weights("r144x72", "r48x24")
    S, D = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
    op = SparseOperator(S, D, w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values, device=0)
    x = field(rng, 12, S, nan_frac=0.01)
    dx = to_device(x)
    dy = DeviceArray((12, D), np.float64)
    s = Stream()
    op.apply(dx, y=dy, stream=s)                       # warm-up outside the capture
    s.synchronize()
    dy.fill_bytes(0)
    graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
    assert hiprt.hipStreamBeginCapture(s.handle, 0) == 0          # hipStreamCaptureModeGlobal
    op.apply(dx, y=dy, stream=s)
    assert hiprt.hipStreamEndCapture(s.handle, ctypes.byref(graph)) == 0
    assert hiprt.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, ctypes.c_size_t(0)) == 0
    for _ in range(5):
        assert hiprt.hipGraphLaunch(gexec, s.handle) == 0
    s.synchronize()
    assert_same(dy.to_host(), oracle.apply_c(op.export_csr(), x), exact=True)
    hiprt.hipGraphExecDestroy(gexec)
    hiprt.hipGraphDestroy(graph)


def test_group_apply_sb_is_graph_capturable(hip, rng):
    """smm_group_apply_sb is one grouped kernel launch on the caller's stream (round 5; round 4 forked per-level
    launches onto a pool of streams): captured it is a single kernel node (the CSR copies exist after the warm-up
    call, so the capture allocates nothing)."""
    from smmregrid_amd import OperatorGroup
    from tests.helpers import ragged_links, random_links
    hiprt = ctypes.CDLL("libamdhip64.so.7")
    S, D, n_ops, B = 800, 190, 5, 24
    ops, csrs = [], []
    imask = (rng.random((n_ops, D)) > 0.3).astype(np.int32)
    frac = rng.random((n_ops, D))
    for i in range(n_ops):
        src, dst, w = (random_links(rng, S, D, 1800) if i % 2 else ragged_links(rng, S, D, max_len=20))
        op = SparseOperator(S, D, src, dst, w, device=0)
        op.set_epilogue(imask[i], frac[i])
        ops.append(op)
        csrs.append(op.export_csr())
    grp = OperatorGroup(ops)
    level_index = np.array([0, 1, 2, 3, 4, 2, 0, 4, 1, 3, 3], dtype=np.int32)
    ml = np.ones(n_ops, np.uint8)
    L = level_index.size
    x = field(rng, B * L, S, nan_frac=0.02).reshape(B, L, 1, S)
    ref = oracle.apply_levels(csrs, x, 1, level_index, ml.astype(bool), imask, frac, 0.4, True)
    xd = to_device(np.ascontiguousarray(np.transpose(x[:, :, 0, :], (1, 2, 0))))
    dy = DeviceArray((B, L, D), np.float64)
    s = Stream()
    grp.apply_sb(xd, level_index, ml, y=dy, masked=True, remap_area_min=0.4, stream=s)   # warm-up: CSR copies
    s.synchronize()
    dy.fill_bytes(0)
    graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
    assert hiprt.hipStreamBeginCapture(s.handle, 0) == 0
    grp.apply_sb(xd, level_index, ml, y=dy, masked=True, remap_area_min=0.4, stream=s)
    assert hiprt.hipStreamEndCapture(s.handle, ctypes.byref(graph)) == 0
    assert hiprt.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, ctypes.c_size_t(0)) == 0
    for _ in range(3):
        assert hiprt.hipGraphLaunch(gexec, s.handle) == 0
    s.synchronize()
    assert_same(dy.to_host().reshape(ref.shape), ref, exact=True)
    hiprt.hipGraphExecDestroy(gexec)
    hiprt.hipGraphDestroy(graph)
    grp.close()
    for op in ops:
        op.close()
