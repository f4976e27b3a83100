"""Handles release what they own: HBM usage returns to its level after operators, groups and
their cached host-pipeline buffers are destroyed."""
import gc

import pytest

from smmregrid_amd import OperatorGroup, SparseOperator, gridgen, to_device
from smmregrid_amd.device import mem_info, synchronize
from tests.helpers import field

pytestmark = pytest.mark.gpu


def test_no_device_memory_leak_over_handle_lifecycles(hip, rng):
    w = gridgen.conservative_weights("r144x72", "r48x24")
    S, D = w.sizes["src_grid_size"], w.sizes["dst_grid_size"]
    x = field(rng, 8, S)

    def cycle():
        ops = [SparseOperator(S, D, w["src_address"].values, w["dst_address"].values, w["remap_matrix"].values,
                              device=0) for _ in range(3)]
        for op in ops:
            op.set_epilogue(w["dst_grid_imask"].values, w["dst_grid_frac"].values)
        grp = OperatorGroup(ops)
        dx = to_device(x)
        ops[0].apply(dx, remap_area_min=0.5).free()
        ops[0].apply_host(x, remap_area_min=0.5)                       # allocates the cached pipeline
        grp.apply_host(x.reshape(2, 2, 2, S), [0, 1], masked=True, remap_area_min=0.5)
        ops[1].mask_apply(w["src_grid_imask"].values)
        dx.free()
        grp.close()
        for op in ops:
            op.close()

    cycle()                                                            # warm-up: runtime pools settle
    gc.collect()
    synchronize()
    free0, _ = mem_info()
    for _ in range(40):
        cycle()
    gc.collect()
    synchronize()
    free1, _ = mem_info()
    assert free0 - free1 < (8 << 20), f"device memory shrank by {(free0 - free1) / 2**20:.1f} MiB over 40 cycles"
