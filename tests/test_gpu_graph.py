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
