"""Tile kernel with LDS-DMA staging (global_load_lds_dwordx4 into a two-slot ring; the default for small
tiles of aligned fields, forced by the tuning knob tile_staging = 2, switched off by 1): bit-identical to the
register-staged kernel and to the oracle."""
import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import SparseOperator, _lib, gridgen, to_device
from tests.helpers import assert_same, field

pytestmark = pytest.mark.gpu
T = _lib.APPLY_KERNEL_TILE
REG, DMA = _lib.STAGING_REGISTERS, _lib.STAGING_DMA
# staging form x rows per step x walk length (smm_debug_set_tuning knobs; {} = the library's own choice)
KNOBS = [{}, dict(tile_staging=DMA, tile_rows_per_step=1), dict(tile_staging=DMA, tile_rows_per_step=1, tile_walk=3),
         dict(tile_staging=REG), dict(tile_staging=REG, tile_rows_per_step=1), dict(tile_staging=DMA, tile_rows_per_step=2),
         dict(tile_staging=DMA, tile_rows_per_step=4), dict(tile_staging=DMA, tile_rows_per_step=2, tile_walk=3),
         dict(tile_staging=DMA, tile_rows_per_step=4, tile_walk=5)]


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
        for knobs in KNOBS:
            with _lib.tuning(**knobs):
                y = op.apply(to_device(x), masked=True, remap_area_min=0.5, flags=T).to_host()
            assert_same(y, ref, exact=True)
