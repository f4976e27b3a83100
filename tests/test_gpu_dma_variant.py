"""Tile kernel with LDS-DMA staging (global_load_lds_dwordx4 into a two-slot ring; the default for small
tiles of aligned fields, forced by tuning variant 10, switched off by 8 / 12): bit-identical to the
register-staged kernel and to the oracle."""
import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import SparseOperator, _lib, gridgen, to_device
from tests.helpers import assert_same, field

pytestmark = pytest.mark.gpu
T, DMA = _lib.APPLY_KERNEL_TILE, _lib.APPLY_KERNEL_TILE | (10 << 16)


@pytest.mark.parametrize("method,src,dst", [("bil", "r360x180", "r90x45"), ("con", "r360x180", "r180x90"),
                                            ("con", "r720x360", "r120x60"), ("bil", "r512x256", "hp32"),
                                            ("con", "r1440x720", "r360x180")])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_dma_staging_matches_oracle(hip, rng, method, src, dst, dtype):
    w = gridgen.generate_weights(src, dst, method=method)
    op = SparseOperator(w.sizes["src_grid_size"], w.sizes["dst_grid_size"], w["src_address"].values,
                        w["dst_address"].values, w["remap_matrix"].values, device=0)
    imask = (rng.random(op.n_dst) > 0.1).astype(np.int32)
    op.set_epilogue(imask, w["dst_grid_frac"].values)
    if not op.plan_info()["tile_plan"]:
        pytest.skip("no tile plan")
    for n_batch in (1, 5, 70):
        x = field(rng, n_batch, op.n_src, dtype=dtype, nan_frac=0.02, inf_frac=0.003)
        ref = oracle.apply_c(op.export_csr(), x, True, imask, w["dst_grid_frac"].values, 0.5)
        for fl in (T, DMA, DMA | (3 << 20), T | (8 << 16), T | (12 << 16), T | (9 << 16), T | (11 << 16),
                   T | (9 << 16) | (3 << 20), T | (11 << 16) | (5 << 20)):
            y = op.apply(to_device(x), masked=True, remap_area_min=0.5, flags=fl).to_host()
            assert_same(y, ref, exact=True)
