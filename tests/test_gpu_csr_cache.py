"""SURVEY f1 'native CSR cache': an operator rebuilt from its exported canonical CSR
(smm_operator_create_csr / SparseOperator.from_csr / save / load) is the same operator."""
import numpy as np
import pytest

from oracle import oracle
from smmregrid_amd import SparseOperator, to_device
from tests.helpers import assert_same, field, ragged_links, random_links

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["random_dups", "ragged"])
def test_csr_roundtrip_is_the_same_operator(hip, rng, tmp_path, kind):
    n_src, n_dst = 4100, 900
    if kind == "random_dups":
        src, dst, w = random_links(rng, n_src, n_dst, 7000, dup_frac=0.2)
    else:
        src, dst, w = ragged_links(rng, n_src, n_dst, max_len=55)
    op = SparseOperator(n_src, n_dst, src, dst, w, device=0)
    imask = (rng.random(n_dst) > 0.2).astype(np.int32)
    frac = rng.random(n_dst)
    op.set_epilogue(imask, frac)
    csr = op.export_csr()
    op2 = SparseOperator.from_csr(n_src, n_dst, *csr, device=0).set_epilogue(imask, frac)
    path = tmp_path / "op.npz"
    op.save(path)
    op3 = SparseOperator.load(path, device=0)
    x = field(rng, 7, n_src, nan_frac=0.02)
    ref = oracle.apply_c(csr, x, True, imask, frac, 0.5)
    for o in (op, op2, op3):
        assert (o.n_src, o.n_dst, o.nnz, o.n_used_src, o.max_row_nnz) == (op.n_src, op.n_dst, op.nnz,
                                                                           op.n_used_src, op.max_row_nnz)
        for a, b in zip(o.export_csr(), csr):
            assert np.array_equal(a, b)
        assert o.plan_info() == op.plan_info()
        assert_same(o.apply(to_device(x), masked=True, remap_area_min=0.5).to_host(), ref, exact=True)


def test_non_canonical_csr_is_rejected(hip):
    rowptr = np.array([0, 2, 3], dtype=np.int64)
    val = np.ones(3)
    with pytest.raises(ValueError, match="ascending"):
        SparseOperator.from_csr(5, 2, rowptr, np.array([3, 1, 0], np.int32), val, device=0)     # unsorted row
    with pytest.raises(ValueError, match="ascending"):
        SparseOperator.from_csr(5, 2, rowptr, np.array([1, 1, 0], np.int32), val, device=0)     # repeated column
    with pytest.raises(ValueError, match="outside"):
        SparseOperator.from_csr(5, 2, rowptr, np.array([1, 5, 0], np.int32), val, device=0)     # column >= n_src
    with pytest.raises(ValueError, match="rowptr"):
        SparseOperator.from_csr(5, 2, np.array([0, 2, 1], np.int64), np.array([1], np.int32), np.ones(1), device=0)
    with pytest.raises(ValueError, match="n_dst"):
        SparseOperator.from_csr(5, 3, rowptr, np.array([0, 1, 2], np.int32), val, device=0)     # rowptr too short


def test_empty_operator_from_csr(hip):
    op = SparseOperator.from_csr(6, 4, np.zeros(5, np.int64), np.zeros(0, np.int32), np.zeros(0), device=0)
    y = op.apply(to_device(np.ones((2, 6)))).to_host()
    assert y.shape == (2, 4) and np.all(y == 0.0)
