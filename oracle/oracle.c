/*
 * oracle.c -- CPU restatement of the smmregrid apply path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's library; the product (smmregrid_amd/) never does.
 *
 * PARITY UNPINNED by reference artefacts: the reference (/root/reference,
 * jhardenberg/smmregrid v0.1.6) cannot be imported here (xarray, dask and
 * sparse are absent) and stores no weights or golden outputs (every numeric
 * test of it needs the cdo binary).  This file therefore restates, step by
 * step, what the reference's own code plus the public semantics of its
 * un-vendored dependencies (pydata/sparse COO, dask.array.tensordot; both
 * unpinned in pyproject.toml:23-33) compute, and is cross-checked in
 * tests/test_oracle.py against an independent numpy/scipy.sparse restatement
 * (oracle/oracle.py), analytic known answers, and outputs of the reference's
 * statement sequence regrid.py:545-570 executed with dask.array on dense weights
 * (tests/golden/dask_statements.npz).
 *
 * Reference lines followed:
 *   weights.py:31-39   src/dst_address - 1, remap_matrix[:,0], COO((src,dst), w, (S,D))
 *                      -> oracle_coo_to_csr (coords sorted, duplicates summed)
 *   weights.py:47-52   mask_tensordot                   -> oracle_mask_apply
 *   regrid.py:545-547  fix_invalid/filled with 1e20     -> fill in oracle_apply
 *   regrid.py:550      tensordot(X(B,S), W(S,D), axes=1)-> oracle_apply
 *   regrid.py:553-570  dst_imask, dst_frac, >1e19       -> epilogue in oracle_apply
 *
 * Arithmetic: per destination cell the links are accumulated sequentially in
 * ascending source index, one rounding for the product and one for the sum
 * (build with -ffp-contract=off), matching a dense x COO loop over
 * lexicographically sorted (src, dst) coordinates.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_F32 0
#define ORACLE_F64 1

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

typedef struct {
  int32_t d, s;
  int64_t k;
} link_t;

static int link_cmp(const void* a, const void* b) {
  const link_t* x = (const link_t*)a;
  const link_t* y = (const link_t*)b;
  if (x->d != y->d) return x->d < y->d ? -1 : 1;
  if (x->s != y->s) return x->s < y->s ? -1 : 1;
  return x->k < y->k ? -1 : (x->k > y->k);
}

/*
 * SCRIP links -> canonical CSR (row = destination, columns ascending,
 * duplicates summed in original link order).  rowptr has n_dst+1 entries;
 * col/val need room for nnz entries; returns the number of entries kept,
 * or -1 on an out-of-range address.
 */
int64_t oracle_coo_to_csr(int64_t n_src, int64_t n_dst, int64_t nnz, const int32_t* src1,
                          const int32_t* dst1, const double* w, int64_t* rowptr, int32_t* col,
                          double* val) {
  link_t* links = (link_t*)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(link_t));
  if (!links) return -2;
  for (int64_t k = 0; k < nnz; ++k) {
    const int64_t s = (int64_t)src1[k] - 1, d = (int64_t)dst1[k] - 1; /* weights.py:31-32 */
    if (s < 0 || s >= n_src || d < 0 || d >= n_dst) {
      free(links);
      return -1;
    }
    links[k].d = (int32_t)d;
    links[k].s = (int32_t)s;
    links[k].k = k;
  }
  qsort(links, (size_t)nnz, sizeof(link_t), link_cmp);
  int64_t n = 0;
  memset(rowptr, 0, (size_t)(n_dst + 1) * sizeof(int64_t));
  for (int64_t i = 0; i < nnz; ++i) {
    if (n > 0 && i > 0 && links[i].d == links[i - 1].d && links[i].s == links[i - 1].s) {
      val[n - 1] += w[links[i].k];
    } else {
      col[n] = links[i].s;
      val[n] = w[links[i].k];
      rowptr[links[i].d + 1]++;
      n++;
    }
  }
  for (int64_t d = 0; d < n_dst; ++d) rowptr[d + 1] += rowptr[d];
  free(links);
  return n;
}

static inline double fetch(const void* x, int dtype, int64_t idx, int fill) {
  if (dtype == ORACLE_F64) {
    double v = ((const double*)x)[idx];
    if (fill && !isfinite(v)) v = 1e20; /* numpy default fill value of float64 */
    return v;
  } else {
    float v = ((const float*)x)[idx];
    if (fill && !isfinite(v)) v = (float)1e20; /* float32(1e20), numpy.ma.fix_invalid on f32 data */
    return (double)v; /* result_type(f32, f64) = f64 */
  }
}

/*
 * Y[b, :] = epilogue( fill(X[b, :]) . W ) for b in [0, n_batch).
 * x rows at x + b*ldx (element units), y (f64) rows at y + b*ldy.
 * dst_imask (int32, nullable) used when masked != 0; dst_frac (nullable) used
 * when area_min > 0.  threads <= 0 -> 1 thread.
 */
void oracle_apply(int64_t n_dst, const int64_t* rowptr, const int32_t* col, const double* val,
                  const void* x, int x_dtype, int64_t ldx, double* y, int64_t ldy,
                  int64_t n_batch, int masked, const int32_t* dst_imask, const double* dst_frac,
                  double area_min, int fill, int threads) {
  (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
#endif
  for (int64_t b = 0; b < n_batch; ++b) {
    const int64_t xo = b * ldx;
    double* yb = y + b * ldy;
    for (int64_t d = 0; d < n_dst; ++d) {
      double acc = 0.0;
      for (int64_t p = rowptr[d]; p < rowptr[d + 1]; ++p) {
        const double prod = val[p] * fetch(x, x_dtype, xo + col[p], fill);
        acc = acc + prod;
      }
      /* regrid.py:553-559 */
      if (masked && dst_imask && dst_imask[d] == 0) acc = NAN;
      /* regrid.py:562-565 */
      if (area_min > 0.0 && dst_frac && dst_frac[d] < area_min) acc = NAN;
      /* regrid.py:570 */
      if (acc > 1e19) acc = NAN;
      yb[d] = acc;
    }
  }
}

/* weights.py:47-52: dst = (src_imask . W) < 0.5 ? 0 : 1 */
void oracle_mask_apply(int64_t n_dst, const int64_t* rowptr, const int32_t* col, const double* val,
                       const int32_t* src_imask, int32_t* dst_imask) {
  for (int64_t d = 0; d < n_dst; ++d) {
    double acc = 0.0;
    for (int64_t p = rowptr[d]; p < rowptr[d + 1]; ++p) {
      const double prod = val[p] * (double)src_imask[col[p]];
      acc = acc + prod;
    }
    dst_imask[d] = acc < 0.5 ? 0 : 1;
  }
}
