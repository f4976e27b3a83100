/* Plain-C consumer of the C ABI (include/smmregrid_amd.h): builds an operator from SCRIP
 * links, regrids a device-resident batch, the same batch through the host pipeline, a two-level
 * group, the batch-fastest entry and a zero-pruned operator fed through a pitched upload, and checks them against a scalar loop.  Compiled with gcc (no HIP headers):
 *   gcc -std=c99 -I include tests/cpp/abi_smoke.c -o abi_smoke smmregrid_amd/libsmmregrid_hip.so -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "smmregrid_amd.h"

#define CHECK(call)                                                        \
  do {                                                                     \
    int rc_ = (call);                                                      \
    if (rc_ != SMM_OK) {                                                   \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, smm_last_error());     \
      return 1;                                                            \
    }                                                                      \
  } while (0)

enum { S = 4096, D = 600, K = 3, B = 5, NNZ = D * K };

int main(void) {
  static int32_t src[NNZ], dst[NNZ];
  static double w[NNZ], x[B * S], y[B * D], yh[B * D], ref[B * D];
  int count = 0;
  CHECK(smm_device_count(&count));
  if (count < 1) {
    fprintf(stderr, "no device\n");
    return 1;
  }
  CHECK(smm_set_device(0));
  for (int d = 0; d < D; ++d)
    for (int k = 0; k < K; ++k) {
      src[d * K + k] = (d * 5 + k * 3) % S + 1; /* 1-based, ascending within a row */
      dst[d * K + k] = d + 1;
      w[d * K + k] = 0.1 + 0.2 * k;
    }
  for (int i = 0; i < B * S; ++i) x[i] = sin(0.001 * i) * 30.0 + 250.0;
  x[7] = NAN; /* becomes 1e20 before the product: weight 0.1 -> finite, others -> NaN */
  for (int b = 0; b < B; ++b)
    for (int d = 0; d < D; ++d) {
      double acc = 0.0;
      for (int k = 0; k < K; ++k) {
        double xv = x[b * S + src[d * K + k] - 1];
        if (!isfinite(xv)) xv = 1e20;
        acc = acc + w[d * K + k] * xv;
      }
      ref[b * D + d] = acc > 1e19 ? NAN : acc;
    }

  smm_operator_t op = NULL;
  CHECK(smm_operator_create(S, D, NNZ, src, dst, w, 0, &op));
  int64_t n_src, n_dst, nnz, used, maxrow;
  CHECK(smm_operator_info(op, &n_src, &n_dst, &nnz, &used, &maxrow));
  if (n_src != S || n_dst != D || nnz != NNZ || maxrow != K) return 2;

  void *dx = NULL, *dy = NULL;
  CHECK(smm_malloc(&dx, sizeof x));
  CHECK(smm_malloc(&dy, sizeof y));
  CHECK(smm_memcpy_h2d(dx, x, sizeof x, NULL));
  CHECK(smm_apply(op, dx, SMM_F64, S, dy, SMM_F64, D, B, 0.0, 0, NULL));
  CHECK(smm_memcpy_d2h(y, dy, sizeof y, NULL));
  CHECK(smm_apply_host(op, x, SMM_F64, S, yh, SMM_F64, D, B, 0.0, 0, 2));

  /* two "levels" sharing the operator: X viewed as (outer = B, lev = 1 ... ) twice */
  smm_operator_t ops[2] = {op, op};
  smm_group_t grp = NULL;
  CHECK(smm_group_create(ops, 2, &grp));
  static double yg[B * D];
  int32_t level_index[1] = {1};
  CHECK(smm_group_apply_host(grp, x, SMM_F64, yg, SMM_F64, B, 1, 1, 1, level_index, NULL, 0.0, 0, 0));

  /* the batch-fastest entry: X transposed to (S, B), Y still (B, D); and an operator created with
   * SMM_CREATE_PRUNE_ZEROS fed through a pitched upload (rows on 128-B lines) -- both must give the same bits */
  static double xt[S * B], ysb[B * D], yp[B * D];
  for (int b = 0; b < B; ++b)
    for (int s = 0; s < S; ++s) xt[s * B + b] = x[b * S + s];
  void* dxt = NULL;
  CHECK(smm_malloc(&dxt, sizeof xt));
  CHECK(smm_memcpy_h2d(dxt, xt, sizeof xt, NULL));
  CHECK(smm_apply_sb(op, dxt, SMM_F64, B, dy, SMM_F64, D, B, 0.0, 0, NULL));
  CHECK(smm_memcpy_d2h(ysb, dy, sizeof ysb, NULL));
  smm_operator_t opp = NULL;
  CHECK(smm_operator_create_opt(S, D, NNZ, src, dst, w, SMM_CREATE_PRUNE_ZEROS, 0, &opp));
  int kind = 0;
  CHECK(smm_operator_plan_info(opp, &kind, NULL, NULL));
  const size_t pitch = ((size_t)S * 8 + 127) / 128 * 128;      /* device rows on 128-B lines */
  void* dxp = NULL;
  CHECK(smm_malloc(&dxp, pitch * B));
  CHECK(smm_memcpy2d_h2d(dxp, pitch, x, (size_t)S * 8, (size_t)S * 8, B, NULL));
  CHECK(smm_apply(opp, dxp, SMM_F64, (int64_t)(pitch / 8), dy, SMM_F64, D, B, 0.0, 0, NULL));
  CHECK(smm_memcpy2d_d2h(yp, (size_t)D * 8, dy, (size_t)D * 8, (size_t)D * 8, B, NULL));
  CHECK(smm_free(dxp));

  int bad = 0;
  for (int i = 0; i < B * D; ++i) {
    const int same = (isnan(ref[i]) && isnan(y[i]) && isnan(yh[i]) && isnan(yg[i]) && isnan(ysb[i]) && isnan(yp[i])) ||
                     (ref[i] == y[i] && ref[i] == yh[i] && ref[i] == yg[i] && ref[i] == ysb[i] && ref[i] == yp[i]);
    if (!same) ++bad;
  }
  /* error path: an address outside the grid is refused with a message */
  int32_t bad_src[1] = {S + 1}, one[1] = {1};
  double ww[1] = {1.0};
  smm_operator_t nope = NULL;
  if (smm_operator_create(S, D, 1, bad_src, one, ww, 0, &nope) != SMM_ERR_INVALID || nope) ++bad;

  CHECK(smm_group_destroy(grp));
  CHECK(smm_operator_destroy(op));
  CHECK(smm_operator_destroy(opp));
  CHECK(smm_free(dxt));
  CHECK(smm_free(dx));
  CHECK(smm_free(dy));
  printf("abi-smoke mismatches=%d last_error=\"%s\"\n", bad, smm_last_error());
  return bad ? 3 : 0;
}
