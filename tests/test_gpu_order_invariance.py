"""What replaces the missing reference pin of the summation order.

The reference builds its matrix with `sparse.COO([src, dst], w)` (weights.py:37-39) and
contracts with `dask.array.tensordot` (regrid.py:550); both live in un-vendored, unpinned
third-party code that cannot run here, so neither the order in which duplicate links are summed
nor the order of the product's accumulation is pinned by any artefact.  These tests show that
the result does not depend on them at the north-star tolerance: the same matrix, written as N
differently ordered and differently split link lists, gives outputs that agree to <= 1e-12
relative with identical NaN positions -- six orders of magnitude inside the 1e-6 the north star
allows -- on the HIP path (both kernels) and against the scipy restatement, whose accumulation
order is a third, independent one.
"""
import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import SparseOperator, _lib, gridgen, to_device
from tests.helpers import field, max_rel_spread, reorder_links

pytestmark = pytest.mark.gpu
SPREAD = 1e-12
N_ORDERS = 8


def _cases(rng):
    w = gridgen.conservative_weights("r144x72", "r48x24")
    yield ("con r144x72->r48x24", 144 * 72, 48 * 24, w["src_address"].values, w["dst_address"].values,
           w["remap_matrix"].values[:, 0], w["dst_grid_frac"].values)
    w = gridgen.bilinear_weights("r180x90", "hp16")
    yield ("bil r180x90->hp16", 180 * 90, 12 * 16 * 16, w["src_address"].values, w["dst_address"].values,
           w["remap_matrix"].values[:, 0], None)
    n_src, n_dst, nnz = 5000, 1200, 30000                     # unstructured, rows of ~25 links
    src = rng.integers(1, n_src + 1, size=nnz).astype(np.int32)
    dst = rng.integers(1, n_dst + 1, size=nnz).astype(np.int32)
    yield ("random positive", n_src, n_dst, src, dst, rng.random(nnz), None)


@pytest.mark.parametrize("kflag", [_lib.APPLY_KERNEL_SELL, 0])
def test_hip_output_independent_of_link_order(hip, rng, kflag):
    for name, n_src, n_dst, src, dst, w, frac in _cases(rng):
        x = field(rng, 9, n_src, nan_frac=0.01, inf_frac=0.002)
        outs, vals = [], []
        for k in range(N_ORDERS):
            s2, d2, w2 = (src, dst, w) if k == 0 else reorder_links(rng, src, dst, w)
            op = SparseOperator(n_src, n_dst, s2, d2, w2, device=0)
            if frac is not None:
                op.set_epilogue(None, frac)
            rowptr, col, val = op.export_csr()
            if k == 0:
                base = (rowptr, col)
            # the integer structure does not depend on the order at all
            assert np.array_equal(rowptr, base[0]) and np.array_equal(col, base[1]), name
            vals.append(val)
            outs.append(op.apply(to_device(x), remap_area_min=0.5 if frac is not None else 0.0,
                                 flags=kflag).to_host())
        assert max_rel_spread(vals) <= 8 * np.finfo(np.float64).eps, name   # duplicate sums: rounding only
        assert max_rel_spread(outs) <= SPREAD, name
        # and a third accumulation order (scipy's CSR product) lands in the same band
        ref = oracle.apply((base[0], base[1], vals[0]), x, False, None, frac, 0.5 if frac is not None else 0.0)
        assert max_rel_spread([outs[0], ref]) <= SPREAD, name
