/*
 * smmregrid_amd.h -- C ABI of libsmmregrid_hip.so (MI355X / gfx950).
 *
 * This library replaces the third-party arithmetic that the reference
 * (jhardenberg/smmregrid) invokes on its hot path -- the seam is
 *
 *   weights.py:25-44   compute_weights_matrix     sparse.COO([src,dst], w, (S,D))
 *   weights.py:7-23    compute_weights_matrix3d   one COO per level, links[:link_length]
 *   weights.py:47-52   mask_tensordot             (src_imask . W) < 0.5 ? 0 : 1
 *   regrid.py:545-547  non-finite -> 1e20 fill of the source array
 *   regrid.py:550      dask.array.tensordot(X(B,S), W(S,D), axes=1)
 *   regrid.py:553-570  where(dst_imask) / where(frac < remap_area_min) / where(> 1e19)
 *   regrid.py:387-418  regrid3d level loop + concat (+ transpose)
 *
 * The reference is pure Python: a maintainer binds this ABI with ctypes
 * (see INTEGRATION.md for the stub). No torch / C++ types cross the boundary:
 * plain pointers, sizes and opaque handles only.
 *
 * Conventions
 *   - every function returns an int status (SMM_OK == 0); it never throws.
 *     The message of the last failure on the calling thread is returned by
 *     smm_last_error().
 *   - "host" pointers are ordinary process memory, borrowed for the call only.
 *     "device" pointers are HBM addresses (from smm_malloc or any hipMalloc).
 *   - operator handles are immutable once their epilogue vectors are set;
 *     smm_apply* is re-entrant on them (the reference's dask scheduler calls
 *     one matrix from several threads, regrid.py:29-30).
 *   - there is NO CPU fallback in this library: without a HIP device every
 *     compute entry point fails with SMM_ERR_NO_DEVICE.
 */
#ifndef SMMREGRID_AMD_H
#define SMMREGRID_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMM_ABI_VERSION 6

/* status codes */
enum {
  SMM_OK = 0,
  SMM_ERR_INVALID = 1,    /* bad argument (address out of range, negative size, ...) */
  SMM_ERR_NO_DEVICE = 2,  /* no HIP device / device ordinal out of range            */
  SMM_ERR_HIP = 3,        /* a HIP runtime call failed (message has the hipError)    */
  SMM_ERR_ALLOC = 4,      /* host allocation failed                                  */
  SMM_ERR_UNSUPPORTED = 5,/* dtype / flag combination not built                      */
  SMM_ERR_INTERNAL = 6    /* an unexpected failure inside the library (message says what) */
};

/* element types of the dense field buffers */
enum {
  SMM_F32 = 0,
  SMM_F64 = 1
};

/* smm_apply flags.  Bits outside this set are refused with SMM_ERR_INVALID by every entry that takes `flags`
 * (ABI <= 4 encoded kernel variants in bits 16..23: those are smm_debug_set_tuning knobs now). */
enum {
  SMM_APPLY_MASKED = 1u << 0,   /* apply dst_imask (regrid.py:553-559); per level in a group */
  SMM_APPLY_NO_FILL = 1u << 1,  /* skip the 1e20 fill: the caller guarantees finite X (results are undefined otherwise) */
  SMM_APPLY_SB_PACKED = 1u << 2, /* smm_apply_sb: X holds only the used source cells (smm_operator_used_sources order) */
  SMM_APPLY_HOST_NO_PACK = 1u << 3, /* smm_apply_host: always ship whole rows (no packing of the used source cells) */
  SMM_APPLY_SB_Y_SB = 1u << 4,   /* smm_apply_sb: the result is kept batch-fastest too, Y (n_dst, ldy >= n_batch) */
  SMM_APPLY_KERNEL_SELL = 1u << 8, /* force the row-per-lane SELL-64 kernel                  */
  SMM_APPLY_KERNEL_TILE = 1u << 9  /* force the LDS-staged source-tile kernel (if planned)   */
};
typedef struct smm_operator* smm_operator_t; /* one (S x D) weights matrix resident in HBM      */
typedef struct smm_group* smm_group_t;       /* ordered set of operators (one per masked level) */

/* ------------------------------------------------------------------ misc */
int smm_abi_version(void);
const char* smm_last_error(void);

/* ------------------------------------------------- device / memory / time */
int smm_device_count(int* count);
int smm_set_device(int device);
int smm_get_device(int* device);
int smm_device_name(int device, char* buf, size_t buflen);
int smm_mem_info(size_t* free_bytes, size_t* total_bytes);
int smm_malloc(void** dptr, size_t bytes);
int smm_free(void* dptr);
int smm_host_alloc(void** hptr, size_t bytes); /* pinned host memory */
int smm_host_free(void* hptr);
/* host -> host copy on the library's staging threads (what the host pipelines use between the caller's arrays and
 * their pinned staging): one thread tops out far below PCIe when it fills a pinned buffer.  The ranges must not overlap. */
int smm_host_memcpy(void* dst_host, const void* src_host, size_t bytes);
int smm_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);
int smm_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
int smm_memcpy_d2d(void* dst_dev, const void* src_dev, size_t bytes, void* stream);
/* pitched copies: `height` rows of `width` BYTES, row r at base + r * pitch (pitches in bytes) -- fields
 * whose device rows start on 128-B lines, columns of a batch-fastest field */
int smm_memcpy2d_h2d(void* dst_dev, size_t dpitch, const void* src_host, size_t spitch, size_t width,
                     size_t height, void* stream);
int smm_memcpy2d_d2h(void* dst_host, size_t dpitch, const void* src_dev, size_t spitch, size_t width,
                     size_t height, void* stream);
int smm_memset(void* dst_dev, int value, size_t bytes, void* stream);
int smm_stream_create(void** stream);
int smm_stream_destroy(void* stream);
int smm_stream_sync(void* stream); /* NULL = default stream */
int smm_device_sync(void);
int smm_event_create(void** event);
int smm_event_destroy(void* event);
int smm_event_record(void* event, void* stream);
int smm_event_sync(void* event);
int smm_stream_wait_event(void* stream, void* event); /* work queued on stream after this call waits for event */
int smm_event_elapsed_ms(void* start, void* stop, float* ms);

/* Synthetic field generator for benchmarks and full-size tests: fills n elements
 * with a counter-based pseudo-normal sequence mean + sigma * N(0,1) (element i
 * depends only on (seed, i), so shards can be filled independently). */
int smm_fill_random(void* dst_dev, int dtype, int64_t n, uint64_t seed, double mean,
                    double sigma, void* stream);

/* ------------------------------------------------------------- operators */

/*
 * Build the operator from SCRIP/CDO links (replaces weights.py:31-42).
 *   src_addr_1based / dst_addr_1based : int32[nnz], 1-based as in the CDO file
 *   w                                  : double[nnz] = remap_matrix[:, 0]
 * Links are sorted by (dst, src) with the original order as tie-break and
 * duplicate (dst, src) pairs are summed in that order (the COO constructor's
 * semantics).  Explicit zero weights are kept.  The handle owns a device copy
 * on `device`; host arrays are only read during the call.
 */
int smm_operator_create(int64_t n_src, int64_t n_dst, int64_t nnz,
                        const int32_t* src_addr_1based,
                        const int32_t* dst_addr_1based,
                        const double* w, int device, smm_operator_t* out);
/*
 * Build the operator from a canonical CSR kept from an earlier smm_operator_export_csr (SURVEY
 * f1 "native CSR cache": skips the sort + duplicate pass of weights.py:37-39 on reload).  rowptr
 * int64[n_dst+1] from 0, col int32 0-based strictly ascending inside a row, val double; anything
 * else is SMM_ERR_INVALID.  Epilogue arrays are set separately as for smm_operator_create.
 */
int smm_operator_create_csr(int64_t n_src, int64_t n_dst, const int64_t* rowptr,
                            const int32_t* col, const double* val, int device,
                            smm_operator_t* out);
/*
 * smm_operator_create with options.
 *   SMM_CREATE_PRUNE_ZEROS  drop links whose (duplicate-summed) weight is exactly zero.  The reference
 *     multiplies them (`sparse.COO` keeps explicit zeros, weights.py:37-39); with the 1e20 fill every
 *     gathered value is finite, so such a link adds +-0.0 to a sum that starts at +0.0 and the results
 *     are bit-identical without it.  Bilinear weights between aligned grids are mostly zeros
 *     (r1440x721 -> r360x180: 2 links of 4).  smm_operator_info / export_csr then describe the pruned
 *     matrix.  Off by default: the operator holds exactly the links it was given.
 */
enum { SMM_CREATE_PRUNE_ZEROS = 1u << 0 };
int smm_operator_create_opt(int64_t n_src, int64_t n_dst, int64_t nnz,
                            const int32_t* src_addr_1based, const int32_t* dst_addr_1based,
                            const double* w, unsigned options, int device, smm_operator_t* out);
int smm_operator_destroy(smm_operator_t op);

/* sizes after duplicate-summing; n_used_src = distinct source cells with >= 1 link (U) */
int smm_operator_info(smm_operator_t op, int64_t* n_src, int64_t* n_dst,
                      int64_t* nnz, int64_t* n_used_src, int64_t* max_row_nnz);

/* canonical CSR (row = destination cell, 0-based, columns ascending) for parity tests.
 * rowptr: int64[n_dst+1], col: int32[nnz], val: double[nnz] (host buffers). */
int smm_operator_export_csr(smm_operator_t op, int64_t* rowptr, int32_t* col, double* val);

/*
 * Epilogue vectors of the weights file (host pointers, copied to HBM):
 *   dst_imask: int32[n_dst] (regrid.py:510, used when SMM_APPLY_MASKED)
 *   dst_frac : double[n_dst] (regrid.py:509, used when area_min > 0)
 * Either may be NULL to clear it.  Call before any concurrent smm_apply and before the operator
 * joins a group (smm_group_create copies the device pointers into the group's level table):
 * SMM_ERR_INVALID while the operator belongs to a group.  On failure the old vectors stay in force.
 */
int smm_operator_set_epilogue(smm_operator_t op, const int32_t* dst_imask,
                              const double* dst_frac);

/*
 * Destination-mask pre-compute (weights.py:47-52):
 *   dst_imask[d] = (sum_s src_imask[s] * W[s,d]) < 0.5 ? 0 : 1
 * src_imask int32[n_src] host, dst_imask int32[n_dst] host (output).
 * Runs on the device (SpMV + threshold).
 */
int smm_operator_mask_apply(smm_operator_t op, const int32_t* src_imask, int32_t* dst_imask);

/* kernel selection the library made for this operator -- kernel_kind bit 0: an LDS tile plan
 * exists, bit 1: it is the default kernel (else SELL row-per-lane), bits 8..: destination rows
 * per block of the plan (256, 64, or 32 / 16 / 8 for rows with very wide footprints) */
int smm_operator_plan_info(smm_operator_t op, int* kernel_kind, int64_t* lds_bytes,
                           int64_t* staged_src_elems);

/* Launch geometry smm_apply would use for n_batch rows of x_dtype under `flags` (nothing is
 * launched; for tests and tuning).  kernel: 0 = SELL row-per-lane, 1 = LDS tile staged through registers,
 * 2 = LDS tile staged by LDS-DMA into a ring of two slots (small tiles of 16-B aligned fields); j_per_block: batch
 * rows walked by one workgroup; rows_per_step: batch rows staged per barrier pair; rows_per_block:
 * destination rows per workgroup; n_blocks: grid size; lds_bytes: dynamic LDS per workgroup;
 * big_operator: the links do not stay in L2, walks are lengthened to amortise their re-read.
 * The answer assumes a field whose base, row pitch and strides are multiples of 16 B (what the LDS-DMA
 * staging needs): smm_apply checks the real field and falls back from kernel 2 to kernel 1 (with that
 * kernel's rows_per_step / lds_bytes) when it is not.  n_blocks is the whole batch; beyond 2^31 - 1
 * workgroups smm_apply launches it in parts.  Any out pointer may be NULL. */
int smm_operator_launch_info(smm_operator_t op, int x_dtype, int64_t n_batch, unsigned flags,
                             int* kernel, int* j_per_block, int* rows_per_step, int* rows_per_block,
                             int64_t* n_blocks, int64_t* lds_bytes, int* big_operator);

/* per-level set (replaces the list built by weights.py:7-23); borrows the operators: they must
 * outlive the group (smm_operator_destroy fails with SMM_ERR_INVALID on a member) */
int smm_group_create(const smm_operator_t* ops, int n_ops, smm_group_t* out);
int smm_group_destroy(smm_group_t g);
/* Uploads the device copy of one (level_index, masked_levels) configuration ahead of time, so
 * that smm_group_apply calls with the same configuration allocate nothing and never block
 * (stream-capturable).  Configurations stay cached until smm_group_destroy (n_lev * 4 + n_ops
 * bytes each); a first-seen configuration passed straight to smm_group_apply is uploaded there
 * (one small hipMalloc + blocking copy, no device synchronisation). */
int smm_group_prepare(smm_group_t g, int64_t n_lev, const int32_t* level_index,
                      const uint8_t* masked_levels);
int smm_group_launch_info(smm_group_t g, int x_dtype, int64_t n_outer, int64_t n_lev, int64_t n_inner,
                          unsigned flags, int* kernel, int* j_per_block, int* rows_per_step,
                          int* rows_per_block, int64_t* n_blocks, int64_t* lds_bytes, int* big_operator);
/* bit 0: every member has an LDS tile plan of the group's block shape, bit 1: the tile kernel is the default */
int smm_group_plan_info(smm_group_t g, int* kernel_kind, int* slices_per_block);

/* ----------------------------------------------------------------- apply */

/*
 * 2-D apply (regrid.py:536-570):   Y[b, :] = epilogue( fill(X[b, :]) . W ),  b in [0, n_batch)
 *   x : device, element type x_dtype, row b starts at x + b*ldx elements (ldx >= n_src)
 *   y : device, element type y_dtype, row b starts at y + b*ldy elements (ldy >= n_dst)
 * The reference always produces f64 (result_type(x, f64)); y_dtype == SMM_F32 is an
 * opt-in narrowing store.
 */
int smm_apply(smm_operator_t op,
              const void* x, int x_dtype, int64_t ldx,
              void* y, int y_dtype, int64_t ldy,
              int64_t n_batch, double remap_area_min, unsigned flags, void* stream);

/*
 * The same product for fields kept BATCH-FASTEST ("SB" layout) by a device-resident producer:
 *   x : device, (n_src, ldx) -- the n_batch values of source cell s are contiguous at x + s*ldx
 *       (ldx >= n_batch); with SMM_APPLY_SB_PACKED x holds only the U used source cells, row r =
 *       the r-th entry of smm_operator_used_sources (ascending source index)
 *   y : device, (n_batch, ldy) exactly as smm_apply writes it (regrid.py:550 layout); with
 *       SMM_APPLY_SB_Y_SB the result stays batch-fastest as well -- y (n_dst, ldy >= n_batch), entry b of
 *       destination cell d at y + d*ldy + b -- which is the x a following smm_apply_sb on the target
 *       grid consumes without any transpose (chains of regrids on device-resident fields)
 * In the reference's native (B, S) layout a stencil that needs 16-B pairs on a 32-B stride (config
 * 2: bilinear 4:1) wastes half of every 128-B line fetched; here every needed source cell is one
 * contiguous run, so HBM traffic equals the algorithmic bytes.  Results are bit-identical to
 * smm_apply on the transposed field.  The first call on an operator uploads its canonical CSR
 * (smm_operator_prepare_sb does that ahead of time); afterwards the call allocates nothing.
 */
int smm_operator_prepare_sb(smm_operator_t op);
/* ascending 0-based indices of the n_used_src source cells that carry a link (host int32[n_used_src]) */
int smm_operator_used_sources(smm_operator_t op, int32_t* used);
int smm_apply_sb(smm_operator_t op,
                 const void* x, int x_dtype, int64_t ldx,
                 void* y, int y_dtype, int64_t ldy,
                 int64_t n_batch, double remap_area_min, unsigned flags, void* stream);

/*
 * Same product for fields in HOST memory (what Regridder.apply_weights receives,
 * regrid.py:537-541): the batch rows are cut into chunks that flow through a
 * double-buffered pipeline -- copy into pinned staging (skipped for pinned
 * buffers, e.g. from smm_host_alloc), H2D, kernel, D2H -- on two streams, so
 * transfers of one chunk overlap the kernel of the other.  Synchronous: Y is
 * complete on return.  chunk_rows <= 0 picks ~256 MiB of X per chunk.
 * This path is PCIe-bound (about 100x below the device-resident rate).  When the operator uses at most
 * four fifths of its source cells (bilinear / nearest downsampling, a masked ocean level) and the batch has >= 8 rows, the staging
 * copy packs only the used cells of a chunk, batch-fastest, and the chunk runs through the kernel of
 * smm_apply_sb: a quarter of the PCIe bytes for config 2, the same bits (SMM_APPLY_HOST_NO_PACK or a
 * forced kernel flag turns it off).
 */
int smm_apply_host(smm_operator_t op,
                   const void* x_host, int x_dtype, int64_t ldx,
                   void* y_host, int y_dtype, int64_t ldy,
                   int64_t n_batch, double remap_area_min, unsigned flags, int64_t chunk_rows);

/*
 * Masked-level apply (regrid.py:387-418 in one launch).  The kept dims of the
 * field are viewed as (n_outer, n_lev, n_inner) around the mask dimension;
 * data level l uses group member level_index[l] (host int32[n_lev], result of
 * the nearest-level match regrid.py:390).  Element offsets:
 *   x row (o,l,i) at  o*xs_outer + l*xs_lev + i*xs_inner
 *   y row (o,l,i) at  o*ys_outer + l*ys_lev + i*ys_inner
 * so both the concat order and the transpose (regrid.py:420-427) are strides.
 * With SMM_APPLY_MASKED, masked_levels (host uint8[n_ops], nullable = all)
 * says per group member whether its dst_imask is applied (regrid.py:405).
 */
int smm_group_apply(smm_group_t g,
                    const void* x, int x_dtype,
                    int64_t xs_outer, int64_t xs_lev, int64_t xs_inner,
                    void* y, int y_dtype,
                    int64_t ys_outer, int64_t ys_lev, int64_t ys_inner,
                    int64_t n_outer, int64_t n_lev, int64_t n_inner,
                    const int32_t* level_index, const uint8_t* masked_levels,
                    double remap_area_min, unsigned flags, void* stream);

/*
 * Masked-level apply for a field kept batch-fastest per level: data level l is an (S, ldx >= n_batch)
 * slab at x + l * xs_lev (the n_batch values of a source cell contiguous), its results go to
 * y + l * ys_lev + b * ys_batch + d.  Y as regrid3d lays it out with transpose (B, L, D):
 * ys_lev = D, ys_batch = L * D; without (L, B, D): ys_lev = B * D, ys_batch = D.  With SMM_APPLY_SB_Y_SB
 * the result stays batch-fastest per level, Y (L, D, ys_batch >= B): ys_lev = D * ys_batch.  Same level_index /
 * masked_levels semantics and the same bits as smm_group_apply.  All data levels run in ONE kernel launch (the
 * levels' CSR pointers travel in the kernel arguments; groups of more than 88 data levels take several launches),
 * ordered on `stream` and capturable into a hipGraph after one warm-up call, which uploads the members' CSR copies
 * (smm_group_prepare_sb does that ahead of time).  BASELINE config 3 kept batch-fastest: 14.4 ms with one launch per
 * level, 9.5 ms grouped.
 */
int smm_group_prepare_sb(smm_group_t g);
int smm_group_apply_sb(smm_group_t g,
                       const void* x, int x_dtype, int64_t xs_lev, int64_t ldx,
                       void* y, int y_dtype, int64_t ys_lev, int64_t ys_batch,
                       int64_t n_batch, int64_t n_lev,
                       const int32_t* level_index, const uint8_t* masked_levels,
                       double remap_area_min, unsigned flags, void* stream);

/*
 * Host-buffer variant (the fields Regridder.regrid3d receives): X host C-contiguous
 * (n_outer, n_lev, n_inner, S); Y host (n_outer, n_inner, n_lev, D) when transpose != 0
 * (regrid.py:420-427) else (n_lev, n_outer, n_inner, D) (regrid.py:410).  Chunks flow through the same
 * double-buffered pipeline as smm_apply_host.  Synchronous.  When the selected levels use at most four fifths of their
 * source cells in total (masked ocean levels thin out with depth) and the batch has >= 8 entries, only the used
 * cells travel over PCIe, packed batch-fastest per level: a chunk is a block of the outer axis with every level when
 * the batch has >= 32 entries and 32 of all levels fit the staging budget, else a few consecutive data levels x a block
 * of the outer axis
 * (BASELINE config 3, 106 GB of X in host memory: 37 GB over PCIe, 0.90 s against 2.04 s for whole rows; same bits).
 * SMM_APPLY_HOST_NO_PACK, a forced kernel flag or a caller's chunk_outer keep whole rows / blocks of the outer axis.
 */
int smm_group_apply_host(smm_group_t g,
                         const void* x_host, int x_dtype, void* y_host, int y_dtype,
                         int64_t n_outer, int64_t n_lev, int64_t n_inner, int transpose,
                         const int32_t* level_index, const uint8_t* masked_levels,
                         double remap_area_min, unsigned flags, int64_t chunk_outer);

/* Test hook of the two host pipelines' error path: chunk number `chunk` (0-based) of every following
 * smm_apply_host / smm_group_apply_host call fails with SMM_ERR_HIP before its copies are queued;
 * chunk < 0 (the initial state) switches it off.  Process-wide; for tests only. */
int smm_debug_fail_at_chunk(int64_t chunk);
/* Test hook of the staging pool behind the host pipelines (one persistent set of worker threads per process, started
 * on first need): no_threads != 0 = behave as if no worker thread could be started (the calling thread then does the
 * staging alone: same bits); throw_in_task >= 0 = the staging task started after that many others throws
 * std::bad_alloc, which the pipeline must turn into SMM_ERR_ALLOC after draining the copies in flight; -1 = off.
 * Process-wide; for tests only. */
int smm_debug_staging_faults(int no_threads, int64_t throw_in_task);

/* Where the host pipelines (smm_apply_host / smm_group_apply_host) spent their time, summed over the calls of this
 * process since the last reset: out[i] for i < n receives entry i of the list below (ms unless said otherwise).
 * STAGE_IN = pack / copy into the pinned staging (calling thread + the staging pool), COPY_OUT = pinned staging ->
 * the caller's Y, WAIT = the calling thread blocked on a chunk's stream; H2D / KERNEL / D2H = per chunk from HIP
 * events on the chunk's stream, summed (the two streams overlap, so the sums may exceed TOTAL, the wall time of the
 * calls).  reset != 0 zeroes the sums afterwards.  For benchmarks and tests. */
enum {
  SMM_HOST_STAT_CALLS = 0, /* count */
  SMM_HOST_STAT_CHUNKS,    /* count */
  SMM_HOST_STAT_STAGE_IN_MS,
  SMM_HOST_STAT_H2D_MS,
  SMM_HOST_STAT_KERNEL_MS,
  SMM_HOST_STAT_D2H_MS,
  SMM_HOST_STAT_COPY_OUT_MS,
  SMM_HOST_STAT_WAIT_MS,
  SMM_HOST_STAT_TOTAL_MS,
  SMM_HOST_STAT_THREADS,   /* staging threads in force at the last call (count) */
  SMM_HOST_STAT_COUNT
};
int smm_debug_host_stats(double* out, int n, int reset);

/* Launch grids are 1-D: a batch whose grid would exceed 2^31 - 1 workgroups is cut into parts that are
 * launched one after the other on the same stream (smm_apply / smm_group_apply: halves of the outer batch
 * range, then of the inner one; smm_apply_sb: runs of whole 128-entry batch tiles).  This test hook lowers
 * that limit so that small inputs reach the split path; 0 restores the default.  Process-wide; for tests only. */
int smm_debug_set_grid_limit(int64_t max_blocks);

/* Tuning knobs of tests, tools and benchmarks -- NOT part of the apply flags, product callers never set them.
 * Every knob only changes how a launch is shaped (which kernel form, how many rows per workgroup, which stream);
 * the results are bit-identical for every value.  value 0 restores the library's own choice; *previous (may be
 * NULL) receives the former value.  Process-wide; set them while no apply call is in flight. */
enum {
  SMM_TUNE_SELL_BATCH_ROWS = 0, /* row-per-lane kernel: batch rows per thread (2, 4, 8)                         */
  SMM_TUNE_TILE_WALK,           /* tile kernel: batch rows walked by one workgroup                             */
  SMM_TUNE_TILE_STAGING,        /* tile kernel: 1 = source tiles staged through registers, 2 = by LDS-DMA      */
  SMM_TUNE_TILE_ROWS_PER_STEP,  /* tile kernel, small tiles: batch rows staged per step (1, 2, 4)              */
  SMM_TUNE_TILE_X_LOADS,        /* tile kernel: 1 = non-temporal, 2 = cached loads of X                        */
  SMM_TUNE_TILE_SPLIT_ROWS,     /* tile kernel, part-of-a-slice blocks: 1 = rows are not split over lane groups */
  SMM_TUNE_TILE_LINKS,          /* single-wave tile kernel: 1 = links streamed per batch row, not kept in registers */
  SMM_TUNE_XCD_RUN,             /* tile + batch-fastest kernels: consecutive blocks per XCD (-1 = dispatcher order) */
  SMM_TUNE_SB_STRIP,            /* batch-fastest kernel: destination tiles per strip (-1 = whole-grid order)   */
  SMM_TUNE_SB_LOADS,            /* batch-fastest kernel: loads per batch of the link walk (4, 8)               */
  SMM_TUNE_SB_LEVEL_LAUNCHES,   /* smm_group_apply_sb: 1 = one launch per data level instead of one grouped launch */
  SMM_TUNE_SB_LDS_PAD,          /* batch-fastest kernels: extra LDS bytes per wave, capping the waves per CU    */
  SMM_TUNE_HOST_PACK_STORES,    /* host pipelines: 1 = the pack writes its staging block with plain (not non-temporal) stores */
  SMM_TUNE_HOST_CHUNK_KB,       /* smm_group_apply_host: > 0 forces level-major packed chunks with this staging budget in KiB (default: 256 MiB, only when a block of the outer axis with all levels does not fit) */
  SMM_TUNE_COUNT
};
int smm_debug_set_tuning(int knob, int value, int* previous);

/* Host threads of operator creation (the sort / duplicate sum replacing weights.py:25-44, the SELL layout and
 * the tile plans are built on several cores) and of the host pipelines' staging copies (one persistent worker
 * pool per process): n > 0 fixes the count, 0 (the initial state) = automatic -- the CPUs the process may really
 * use (scheduler affinity capped by the cgroup CPU quota), at most 16, creations running at the same moment
 * sharing them.  Results do not depend on the count.  *previous (may be NULL) receives the former setting.
 * Process-wide. */
int smm_set_host_threads(int n, int* previous);

/* ------------------------------------------------- multi-GPU exchange (RCCL over xGMI) */

/*
 * One process per GPU; batch rows are sharded over the ranks with no collective on the data
 * path; these calls move the Y shards afterwards.  Rank 0 obtains an id with
 * smm_comm_unique_id, distributes its SMM_COMM_ID_BYTES bytes to the other ranks by any
 * host channel, then every rank (after smm_set_device) calls smm_comm_create.  librccl is
 * bound at run time; SMM_ERR_UNSUPPORTED if it cannot be loaded.
 *   gather   : root's recv_dev holds n_ranks * count elements, rank r's shard at r * count
 *   allgather: every rank's recv_dev holds n_ranks * count elements
 * Both are asynchronous on `stream`.
 */
#define SMM_COMM_ID_BYTES 128
typedef struct smm_comm* smm_comm_t;
int smm_comm_unique_id(void* id_out);
int smm_comm_create(const void* id, int n_ranks, int rank, smm_comm_t* out);
int smm_comm_destroy(smm_comm_t c);
int smm_comm_gather(smm_comm_t c, const void* send_dev, void* recv_dev, int64_t count, int dtype,
                    int root, void* stream);
int smm_comm_allgather(smm_comm_t c, const void* send_dev, void* recv_dev, int64_t count, int dtype,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMMREGRID_AMD_H */
