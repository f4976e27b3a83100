"""Golden vectors from the reference's own third-party calls (dask.array).

Run with an interpreter that has dask (here: /opt/conda/bin/python3.9, dask 2021.10):

    /opt/conda/bin/python3.9 tests/golden/make_dask_golden.py

The reference package itself cannot be imported in this environment (xarray and
sparse are absent), but the statement sequence of regrid.py:545-570 only needs
dask.array -- with the sparse matrix replaced by its dense equivalent it is
executed here verbatim: ma.set_fill_value / ma.fix_invalid / ma.filled, tensordot,
and the three `where`s.  This pins what those dask/numpy calls really do (the
fill value of a float32 field, the treatment of +-inf, the NaN logic) for the
oracle; the summation order of a dense BLAS product differs from the sparse
loop, so values are compared to 1e-12, NaN positions exactly.
"""
import os

import dask
import dask.array
import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
rng = numpy.random.default_rng(20260723)


def reference_statements(source_array, weights_dense, masked, dst_grid_mask, dst_frac_area, remap_area_min):
    kept_shape = list(source_array.shape[:-1])
    # --- regrid.py:545-547
    dask.array.ma.set_fill_value(source_array, 1e20)
    source_array = dask.array.ma.fix_invalid(source_array)
    source_array = dask.array.ma.filled(source_array)
    # --- regrid.py:550
    target_dask = dask.array.tensordot(source_array, weights_dense, axes=1)
    # --- regrid.py:553-559
    if masked:
        mask_shape = [1 for d in kept_shape] + [-1]
        target_mask = dst_grid_mask.reshape(mask_shape).astype(bool)
        target_dask = dask.array.where(target_mask, target_dask, numpy.nan)
    # --- regrid.py:562-565
    if remap_area_min > 0.0:
        target_dask = dask.array.where(
            dask.array.broadcast_to(dst_frac_area, target_dask.shape) < remap_area_min,
            numpy.nan, target_dask)
    # --- regrid.py:570
    target_dask = dask.array.where(target_dask > 1e19, numpy.nan, target_dask)
    return numpy.asarray(target_dask.compute())


def main():
    S, D, nnz = 60, 20, 90
    src = rng.integers(1, S + 1, nnz).astype(numpy.int32)
    dst = rng.integers(1, D + 1, nnz).astype(numpy.int32)
    w = rng.uniform(0.0, 1.0, nnz)
    w[:5] = 0.05                                  # small weights: the 1e20 quirk stays finite
    dense = numpy.zeros((S, D))
    numpy.add.at(dense, (src - 1, dst - 1), w)    # COO semantics: duplicates sum
    dst_mask = (rng.random(D) > 0.2).astype(numpy.int32)
    frac = rng.random(D)
    out = {"n_src": S, "n_dst": D, "src_address": src, "dst_address": dst, "remap_matrix": w,
           "dst_imask": dst_mask, "dst_frac": frac}
    for dtype in (numpy.float64, numpy.float32):
        x = (250 + 30 * rng.standard_normal((2, 3, S))).astype(dtype)
        x[0, 1, ::7] = numpy.nan
        x[1, 0, 3] = numpy.inf
        x[1, 2, 5] = -numpy.inf
        tag = numpy.dtype(dtype).name
        out["x_" + tag] = x
        for masked, amin in ((False, 0.0), (True, 0.0), (True, 0.5), (False, 0.9)):
            for lazy in (False, True):
                xa = dask.array.from_array(x, chunks=(1, 3, S)) if lazy else x
                y = reference_statements(xa, dense, masked, dst_mask, frac, amin)
                key = "y_%s_m%d_a%d" % (tag, int(masked), int(amin * 10))
                if lazy:
                    assert numpy.array_equal(numpy.isnan(y), numpy.isnan(out[key]))
                    assert numpy.allclose(y, out[key], rtol=1e-13, equal_nan=True)
                else:
                    out[key] = y
        assert out["y_%s_m0_a0" % tag].dtype == numpy.float64     # result_type(x, f64)
    # --- weights.py:47-52 mask_tensordot, verbatim, with int32 source masks
    for i in range(3):
        src_mask = (rng.random(S) > 0.3 * (i + 1) / 3).astype(numpy.int32)
        target_mask = dask.array.tensordot(src_mask, dense, axes=1)
        target_mask = dask.array.where(target_mask < 0.5, 0, 1)
        out["src_imask_%d" % i] = src_mask
        out["mask_tensordot_%d" % i] = numpy.asarray(target_mask.compute())
    out["versions"] = numpy.array("dask %s, numpy %s" % (dask.__version__, numpy.__version__))
    numpy.savez_compressed(os.path.join(HERE, "dask_statements.npz"), **out)
    print("written", out["versions"])


if __name__ == "__main__":
    main()
