// libsmmregrid_hip: HIP kernels (gfx950) and the C ABI declared in
// include/smmregrid_amd.h.
//
// Hot path of jhardenberg/smmregrid rebuilt for MI355X:
//   regrid.py:545-547  fill of non-finite source values with 1e20     (fused, on load)
//   regrid.py:550      tensordot(X(B,S), W(S,D))                      (CSR SpMM, HBM-bound)
//   regrid.py:553-570  dst_imask / dst_frac / >1e19 -> NaN            (fused, on store)
//   regrid.py:387-418  per-level loop, concat, transpose              (one grouped launch)
//   weights.py:47-52   mask pre-compute                               (same kernel, B = 1)
//
// X keeps the reference's native layout: batch rows of S contiguous source
// cells (regrid.py:539-541), so no transpose pass is ever made.  The product
// is memory bound (2 flop per gathered 8-B element): the kernels are built
// around HBM traffic, not MFMA.
//
// Summation order: links of a destination row are accumulated sequentially in
// ascending source index with separate multiply and add (no FMA contraction),
// exactly the order of the CPU oracle (oracle/), so f64 results are bit
// identical to it.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/smmregrid_amd.h"
#include "smm_internal.h"
#include "smm_launch.hpp"

#pragma clang fp contract(off)

namespace smm_launch {
#define SMM_EXTERN_PAIR(XT, YT)                                                                         \
  extern template int launch_sell<XT, YT>(const ApplyArgs&, int64_t, bool, unsigned, hipStream_t);      \
  extern template int launch_tile<XT, YT>(const ApplyArgs&, int64_t, int, int64_t, int64_t, int, bool, \
                                          unsigned, hipStream_t);                                       \
  extern template int launch_sb<XT, YT>(const SbArgs&, bool, unsigned, hipStream_t);                   \
  extern template int launch_sb_group<XT, YT>(const SbGroupArgs&, bool, unsigned, hipStream_t);
SMM_EXTERN_PAIR(double, double)
SMM_EXTERN_PAIR(double, float)
SMM_EXTERN_PAIR(float, double)
SMM_EXTERN_PAIR(float, float)
#undef SMM_EXTERN_PAIR
}  // namespace smm_launch

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

}  // namespace

namespace {
// smm_debug_set_tuning: process-wide knobs of tests / tools / benchmarks, 0 = the library's own choice
std::atomic<int> g_tuning[SMM_TUNE_COUNT];
}  // namespace

namespace smm {
int fail_msg(int code, const std::string& msg) { return fail(code, msg); }  // for smm_comm.cpp
int tuning(int knob) { return (knob >= 0 && knob < SMM_TUNE_COUNT) ? g_tuning[knob].load(std::memory_order_relaxed) : 0; }
}

namespace {

#define SMM_HIP(call)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) {                                                             \
      (void)hipGetLastError();                                                          \
      return fail(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice                 \
                      ? SMM_ERR_NO_DEVICE                                               \
                      : SMM_ERR_HIP,                                                    \
                  std::string(#call) + ": " + hipGetErrorString(e_));                   \
    }                                                                                   \
  } while (0)

// Staging resources of the host-buffer pipeline, cached per operator (allocation costs
// milliseconds, a small regrid microseconds): two streams, two device X/Y chunk buffers,
// two pinned X/Y staging buffers, grown on demand.
struct HostPipe {
  hipStream_t stream[2] = {nullptr, nullptr};
  void* dx[2] = {nullptr, nullptr};
  void* dy[2] = {nullptr, nullptr};
  void* hx[2] = {nullptr, nullptr};
  void* hy[2] = {nullptr, nullptr};
  size_t cap_dx = 0, cap_dy = 0, cap_hx = 0, cap_hy = 0;
  // per buffer: before the H2D, after it, after the kernel(s), after the D2H -- the stage times of a chunk
  // (smm_debug_host_stats) are read from them once the chunk has been drained
  hipEvent_t ev[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
  hipError_t ensure(size_t need_dx, size_t need_dy, size_t need_hx, size_t need_hy) {
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; ++i)
      if (!stream[i]) e = hipStreamCreateWithFlags(&stream[i], hipStreamNonBlocking);
    for (int i = 0; i < 2; ++i)
      for (int k = 0; k < 4 && e == hipSuccess; ++k)
        if (!ev[i][k]) e = hipEventCreate(&ev[i][k]);
    auto grow_dev = [&](void* (&buf)[2], size_t& cap, size_t need) {
      if (need <= cap || e != hipSuccess) return;
      for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        (void)hipFree(buf[i]);
        buf[i] = nullptr;
        e = hipMalloc(&buf[i], need);
      }
      cap = e == hipSuccess ? need : 0;
    };
    auto grow_host = [&](void* (&buf)[2], size_t& cap, size_t need) {
      if (need <= cap || e != hipSuccess) return;
      for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        if (buf[i]) (void)hipHostFree(buf[i]);
        buf[i] = nullptr;
        e = hipHostMalloc(&buf[i], need, hipHostMallocDefault);
      }
      cap = e == hipSuccess ? need : 0;
    };
    grow_dev(dx, cap_dx, need_dx);
    grow_dev(dy, cap_dy, need_dy);
    grow_host(hx, cap_hx, need_hx);
    grow_host(hy, cap_hy, need_hy);
    return e;
  }
  // After a failed chunk: wait for whatever is still queued on both streams (an async D2H into the
  // caller's Y of the previous chunk), so that nothing writes into caller memory after the return.
  void quiesce() {
    for (int i = 0; i < 2; ++i)
      if (stream[i]) (void)hipStreamSynchronize(stream[i]);
    (void)hipGetLastError();
  }
  ~HostPipe() {
    for (int i = 0; i < 2; ++i) {
      if (stream[i]) (void)hipStreamDestroy(stream[i]);
      for (int k = 0; k < 4; ++k)
        if (ev[i][k]) (void)hipEventDestroy(ev[i][k]);
      (void)hipFree(dx[i]);
      (void)hipFree(dy[i]);
      if (hx[i]) (void)hipHostFree(hx[i]);
      if (hy[i]) (void)hipHostFree(hy[i]);
    }
  }
};

}  // namespace

// ------------------------------------------------------------------ handles

constexpr int kNumShapes = 5;
constexpr int shape_rows(int which) { return which == 0 ? 256 : (64 >> (which - 1)); }

struct smm_operator {
  int device = -1;
  smm::HostCsr csr;          // canonical: row = destination cell
  int64_t pruned_links = 0;  // exact-zero links dropped at create time (SMM_CREATE_PRUNE_ZEROS)
  int64_t n_slices = 0, n_slots = 0;
  int64_t* d_slice_off = nullptr;
  int32_t* d_col = nullptr;
  double* d_val = nullptr;
  int32_t* d_rowlen = nullptr;
  uint8_t* d_imask = nullptr;
  double* d_frac = nullptr;
  // LDS tile plans by block shape: [0] = 4 slices (256 rows) per block, [1] = 1 slice (heavy rows),
  // [2..4] = 32 / 16 / 8 rows of a slice (rows so long -- high-resolution source, coarse target --
  // that a whole slice's footprint exceeds the LDS budget).  The operator's own shape is built at
  // create time, the others on demand when it joins a group of another shape.
  struct TilePlan {
    bool built = false, valid = false;
    int64_t max_chunks = 0, total_chunks = 0, total_lines = 0;
    bool preferred = false;  // staged lines are used well enough to beat direct gathers
    bool reuse = false;      // some staged lines are shared by several blocks (keep them cacheable)
    int64_t* d_blk_chunk_off = nullptr;
    int32_t* d_chunk_src = nullptr;
    int32_t* d_lcol = nullptr;
    uint8_t* d_blk_direct = nullptr;
  } plan[kNumShapes];
  smm::HostSell sell_shape;  // slice_off / rowlen only (col/val dropped after upload)
  std::mutex plan_mu;
  std::mutex pipe_mu;        // smm_apply_host calls on one operator take turns
  HostPipe pipe;
  // plain canonical CSR on the device for the batch-fastest kernel, uploaded on first use
  bool sb_ready = false;
  std::vector<int32_t> h_used;        // ascending used source cells (host pack of the pipeline)
  int64_t* d_csr_rowptr = nullptr;
  int32_t* d_csr_col = nullptr;       // source cell
  int32_t* d_csr_colp = nullptr;      // rank of the source cell among the used cells (packed X)
  double* d_csr_val = nullptr;
  std::atomic<int> group_refs{0};  // groups borrowing this operator (their descriptors hold its device pointers)
  int native = 0;            // shape of the operator's own plan (choose_native_plan)
  int native_plan() const { return native; }
  LevelDesc* d_desc = nullptr;  // one-element device copy (native plan)
  LevelDesc desc(int which) const {
    LevelDesc L;
    L.slice_off = d_slice_off;
    L.col = d_col;
    L.val = d_val;
    L.rowlen = d_rowlen;
    L.imask = d_imask;
    L.frac = d_frac;
    L.blk_chunk_off = plan[which].d_blk_chunk_off;
    L.chunk_src = plan[which].d_chunk_src;
    L.lcol = plan[which].d_lcol;
    L.blk_direct = plan[which].d_blk_direct;
    return L;
  }
};

struct smm_group {
  int device = -1;
  std::vector<smm_operator_t> ops;
  LevelDesc* d_descs = nullptr;
  int tile_which = 0;  // plan shape shared by all members
  bool tile_valid = false;
  bool tile_preferred = false;
  bool tile_reuse = false;
  int64_t tile_max_chunks = 0;
  int64_t max_row_nnz = 0;
  // uploaded (level_index, masked_levels) configurations, keyed by content.  An entry lives until
  // smm_group_destroy: a kernel enqueued by another thread may still read it, so nothing is ever
  // evicted (an entry is n_lev * 4 + n_ops bytes; callers cycle through a few level subsets).
  std::mutex mu;
  std::map<std::string, void*> cfg_cache;
  std::mutex pipe_mu;  // smm_group_apply_host calls on one group take turns
  HostPipe pipe;
};

namespace {

template <typename T>
int upload(T** dptr, const std::vector<T>& h) {
  *dptr = nullptr;
  const size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
  SMM_HIP(hipMalloc((void**)dptr, bytes));
  if (!h.empty()) SMM_HIP(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return SMM_OK;
}

struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    if (prev == dev) {
      ok = true;
      return;
    }
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (ok && prev >= 0) (void)hipSetDevice(prev);
  }
};

int refresh_desc(smm_operator* op) {
  const LevelDesc L = op->desc(op->native_plan());
  if (!op->d_desc) SMM_HIP(hipMalloc((void**)&op->d_desc, sizeof(LevelDesc)));
  SMM_HIP(hipMemcpy(op->d_desc, &L, sizeof(LevelDesc), hipMemcpyHostToDevice));
  return SMM_OK;
}

void release(smm_operator* op) {
  if (!op) return;
  (void)hipFree(op->d_slice_off);
  (void)hipFree(op->d_col);
  (void)hipFree(op->d_val);
  (void)hipFree(op->d_rowlen);
  (void)hipFree(op->d_imask);
  (void)hipFree(op->d_frac);
  for (auto& pl : op->plan) {
    (void)hipFree(pl.d_blk_chunk_off);
    (void)hipFree(pl.d_chunk_src);
    (void)hipFree(pl.d_lcol);
    (void)hipFree(pl.d_blk_direct);
  }
  (void)hipFree(op->d_desc);
  (void)hipFree(op->d_csr_rowptr);
  (void)hipFree(op->d_csr_col);
  (void)hipFree(op->d_csr_colp);
  (void)hipFree(op->d_csr_val);
  delete op;
}

// Build and upload tile plan `which` of the operator if it does not exist yet.
int ensure_plan(smm_operator* op, int which) {
  std::lock_guard<std::mutex> lock(op->plan_mu);
  smm_operator::TilePlan& pl = op->plan[which];
  if (pl.built) return SMM_OK;
  smm::HostTilePlan hp;
  // LDS / staging-register budget: 64 KiB per 4-wave block, 16 KiB per single-wave block (whatever
  // part of the slice's rows it owns)
  const int64_t budget = which == 0 ? kTileMaxChunks : kTileMaxChunks / kWavesPerBlock;
  smm::build_tile_plan(op->csr, op->sell_shape, shape_rows(which), kChunkElems, budget, hp);
  smm::tighten_tile_plan(op->csr, hp, budget);
  pl.built = true;
  if (!hp.valid) return SMM_OK;
  int rc = SMM_OK;
  if ((rc = upload(&pl.d_blk_chunk_off, hp.blk_chunk_off)) || (rc = upload(&pl.d_chunk_src, hp.chunk_src)) ||
      (rc = upload(&pl.d_lcol, hp.lcol)) || (rc = upload(&pl.d_blk_direct, hp.blk_direct)))
    return rc;
  pl.valid = true;
  pl.max_chunks = hp.max_block_chunks;
  pl.total_chunks = hp.total_chunks;
  pl.total_lines = hp.total_lines;
  pl.reuse = hp.total_chunks * 50 > hp.distinct_chunks * 51;  // > 2 % of lines staged twice
  // at least a tenth of every staged 128-B line is consumed: the lines are the ones a gather would
  // fetch anyway, and staging fetches them coalesced (r3600x1800 -> r360x180 bilinear uses 20 %:
  // tile 0.48 ms, SELL 0.64 ms; HEALPix-nested source, nearest neighbour, 14 %: 1.87 vs 2.10 ms;
  // r3600x1800 nearest neighbour, 10 %: equal)
  pl.preferred = hp.total_distinct * 10 >= hp.total_lines * 16;
  return SMM_OK;
}

// launch_sell / launch_tile live in smm_launch.hpp; their four dtype instantiations are compiled in
// smm_launch_inst.hip (one object per pair) and only declared here
using smm_launch::launch_sell;
using smm_launch::launch_tile;
using smm_launch::launch_sb;
using smm_launch::launch_sb_group;

// Device copy of the canonical CSR (+ packed column ranks) for the batch-fastest kernel.
int ensure_sb(smm_operator* op) {
  std::lock_guard<std::mutex> lock(op->plan_mu);
  if (op->sb_ready) return SMM_OK;
  const smm::HostCsr& c = op->csr;
  std::vector<int32_t> rank((size_t)std::max<int64_t>(c.n_src, 1), -1), colp((size_t)c.nnz);
  for (int32_t s : c.col) rank[(size_t)s] = 0;
  int32_t r = 0;
  for (int64_t s = 0; s < c.n_src; ++s)
    if (rank[(size_t)s] == 0) rank[(size_t)s] = r++;
  for (int64_t i = 0; i < c.nnz; ++i) colp[(size_t)i] = rank[(size_t)c.col[(size_t)i]];
  op->h_used.clear();
  op->h_used.reserve((size_t)c.n_used_src);
  for (int64_t s = 0; s < c.n_src; ++s)
    if (rank[(size_t)s] >= 0) op->h_used.push_back((int32_t)s);
  int rc = SMM_OK;
  if ((rc = upload(&op->d_csr_rowptr, c.rowptr)) || (rc = upload(&op->d_csr_col, c.col)) ||
      (rc = upload(&op->d_csr_colp, colp)) || (rc = upload(&op->d_csr_val, c.val))) {
    (void)hipFree(op->d_csr_rowptr);
    (void)hipFree(op->d_csr_col);
    (void)hipFree(op->d_csr_colp);
    (void)hipFree(op->d_csr_val);
    op->d_csr_rowptr = nullptr;
    op->d_csr_col = op->d_csr_colp = nullptr;
    op->d_csr_val = nullptr;
    return rc;
  }
  op->sb_ready = true;
  return SMM_OK;
}

// largest 1-D launch grid (workgroups); smm_debug_set_grid_limit lowers it so that tests reach the split path
std::atomic<int64_t> g_grid_limit{0x7fffffffLL};
inline int64_t grid_limit() { return g_grid_limit.load(); }

struct LaunchInfo {
  bool tile = false, big_operator = false, dma = false;
  int j_per_block = 0, rows_per_step = 1, rows_per_block = 0;
  int64_t n_jtiles = 0, n_blocks = 0, lds_bytes = 0;
};

// Common launch path for a single operator (descs = op->d_desc) or a group.
int run_apply(const LevelDesc* d_descs, const int32_t* d_lev_map, const uint8_t* d_lev_masked,
              int64_t n_src, int64_t n_dst, int tile_which, bool tile_ok, bool tile_preferred,
              int tile_flags, int64_t tile_max_chunks,
              int64_t max_row_nnz, const void* x, int x_dtype, int64_t xs_o, int64_t xs_l,
              int64_t xs_i, void* y, int y_dtype, int64_t ys_o, int64_t ys_l, int64_t ys_i,
              int64_t n_outer, int64_t n_lev, int64_t n_inner, double area_min, unsigned flags,
              hipStream_t s, LaunchInfo* info_only = nullptr) {
  if (n_outer < 0 || n_lev < 0 || n_inner < 0) return fail(SMM_ERR_INVALID, "negative batch size");
  if (n_outer == 0 || n_lev == 0 || n_inner == 0 || n_dst == 0) return SMM_OK;
  if (!info_only && (!x || !y)) return fail(SMM_ERR_INVALID, "null field pointer");
  if ((x_dtype != SMM_F32 && x_dtype != SMM_F64) || (y_dtype != SMM_F32 && y_dtype != SMM_F64))
    return fail(SMM_ERR_UNSUPPORTED, "field dtype must be SMM_F32 or SMM_F64");
  if (!(area_min >= 0.0 && area_min <= 1.0))
    return fail(SMM_ERR_INVALID, "remap_area_min must be within [0, 1]");  // regrid.py:124-125

  ApplyArgs a{};
  a.descs = d_descs;
  a.lev_map = d_lev_map;
  a.lev_masked = d_lev_masked;
  a.x = x;
  a.y = y;
  a.xs_o = xs_o;
  a.xs_l = xs_l;
  a.xs_i = xs_i;
  a.ys_o = ys_o;
  a.ys_l = ys_l;
  a.ys_i = ys_i;
  a.n_j = n_outer * n_inner;
  a.n_inner = n_inner;
  a.n_src = n_src;
  a.n_dst = n_dst;
  a.n_dblocks = ((n_dst + 63) / 64 + kWavesPerBlock - 1) / kWavesPerBlock;  // SELL: 4 slices per workgroup
  a.area_min = area_min;
  a.masked = (flags & SMM_APPLY_MASKED) ? 1 : 0;
  const bool fill = !(flags & SMM_APPLY_NO_FILL);

  const size_t xsz = x_dtype == SMM_F64 ? 8 : 4;
  bool use_tile = false;
  if (flags & SMM_APPLY_KERNEL_TILE) {
    if (!tile_ok) return fail(SMM_ERR_UNSUPPORTED, "operator has no LDS tile plan");
    use_tile = true;
  } else if (!(flags & SMM_APPLY_KERNEL_SELL)) {
    use_tile = tile_ok && tile_preferred;
  }
  // The staging loads are 16 B wide but only need element alignment (unaligned 16-B global loads
  // are legal on gfx950; rows of odd length still run ~10 % faster than the SELL kernel).
  if (use_tile && ((uintptr_t)x % xsz) != 0) {
    if (flags & SMM_APPLY_KERNEL_TILE) return fail(SMM_ERR_INVALID, "field pointer is not element aligned");
    use_tile = false;
  }

#define SMM_DISPATCH(FN, ...)                                                        \
  (x_dtype == SMM_F64                                                                \
       ? (y_dtype == SMM_F64 ? FN<double, double>(__VA_ARGS__) : FN<double, float>(__VA_ARGS__)) \
       : (y_dtype == SMM_F64 ? FN<float, double>(__VA_ARGS__) : FN<float, float>(__VA_ARGS__)))
  if (use_tile) {
    const int64_t rows = shape_rows(tile_which);          // destination rows per block of the tile plan
    a.n_dblocks = (n_dst + rows - 1) / rows;
    a.sub_shift = tile_which >= 2 ? tile_which - 1 : 0;   // block = 1 / 2^sub_shift of a slice
  }
  if (info_only) {
    info_only->tile = use_tile;
    if (use_tile) {
      const smm_launch::TileLaunchCfg c =
          smm_launch::tile_launch_cfg(a, n_lev, tile_which, tile_max_chunks, max_row_nnz, flags, xsz);
      info_only->j_per_block = c.j_per_block;
      info_only->n_jtiles = c.n_jtiles;
      info_only->n_blocks = c.total;
      info_only->rows_per_step = c.rows;
      info_only->lds_bytes = (int64_t)c.lds;
      info_only->big_operator = c.big_operator;
      info_only->dma = c.dma;
      info_only->rows_per_block = (int)shape_rows(tile_which);
    } else {
      const int bt = smm_launch::sell_batch_rows(a.n_j);
      info_only->j_per_block = bt;
      info_only->n_jtiles = (a.n_j + bt - 1) / bt;
      info_only->n_blocks = a.n_dblocks * info_only->n_jtiles * n_lev;
      info_only->rows_per_step = bt;
      info_only->rows_per_block = 256;
    }
    return SMM_OK;
  }
  // One launch when the grid fits (always, short of ~2^31 workgroups); else the batch is cut into parts
  // (smm::split_batch) that are launched one after the other on the same stream.
  auto blocks_for = [&](int64_t n_o, int64_t n_i) -> int64_t {
    ApplyArgs t = a;
    t.n_j = n_o * n_i;
    t.n_inner = n_i;
    if (use_tile)
      return smm_launch::tile_launch_cfg(t, n_lev, tile_which, tile_max_chunks, max_row_nnz, flags, xsz).total;
    const int bt = smm_launch::sell_batch_rows(t.n_j);
    return t.n_dblocks * ((t.n_j + bt - 1) / bt) * n_lev;
  };
  const size_t ysz = y_dtype == SMM_F64 ? 8 : 4;
  auto launch_part = [&](int64_t o0, int64_t n_o, int64_t i0, int64_t n_i) -> int {
    ApplyArgs p = a;
    p.x = (const char*)x + (o0 * xs_o + i0 * xs_i) * (int64_t)xsz;
    p.y = (char*)y + (o0 * ys_o + i0 * ys_i) * (int64_t)ysz;
    p.n_j = n_o * n_i;
    p.n_inner = n_i;
    if (use_tile)
      return SMM_DISPATCH(launch_tile, p, n_lev, tile_which, tile_max_chunks, max_row_nnz, tile_flags, fill, flags, s);
    return SMM_DISPATCH(launch_sell, p, n_lev, fill, flags, s);
  };
  const int rc = smm::split_batch(0, n_outer, 0, n_inner, grid_limit(), blocks_for, launch_part);
  if (rc == -1)
    return fail(SMM_ERR_INVALID, "one batch row alone needs a launch grid beyond " + std::to_string(grid_limit()) +
                                     " workgroups (destination blocks x levels)");
  return rc;
#undef SMM_DISPATCH
}

}  // namespace

namespace {

// test hook for the pipelines' error path (smm_debug_fail_at_chunk): chunk c of the next host-pipeline
// calls fails; -1 (the default) = off.  Set explicitly by the tests, never read from the environment.
std::atomic<int64_t> g_fail_at_chunk{-1};
int64_t test_fail_chunk() { return g_fail_at_chunk.load(std::memory_order_relaxed); }

// Staging stages of the two host pipelines (smm_hostpool.cpp: one persistent worker pool, nothing throws):
// their int results become statuses here.
int stage_status(int rc, const char* what) {
  if (rc == 0) return SMM_OK;
  return fail(rc == 1 ? SMM_ERR_ALLOC : SMM_ERR_INTERNAL,
              std::string(what) + (rc == 1 ? ": out of host memory in a staging task" : ": a staging task failed"));
}
int host_copy(void* dst, const void* src, size_t bytes) { return stage_status(smm::host_copy(dst, src, bytes), "host copy"); }
int host_pack(void* out, const void* x, size_t xsz, int64_t n_inner, int64_t stride_o, int64_t stride_i,
              const std::vector<int32_t>& used, int64_t rows) {
  return stage_status(smm::host_pack(out, x, xsz, n_inner, stride_o, stride_i, used.data(), (int64_t)used.size(), rows,
                                     smm::tuning(SMM_TUNE_HOST_PACK_STORES) != 1),
                      "host pack");
}
int host_pack(void* out, const void* x, size_t xsz, int64_t ldx, const std::vector<int32_t>& used, int64_t rows) {
  return host_pack(out, x, xsz, std::max<int64_t>(rows, 1), 0, ldx, used, rows);
}

// What the two host pipelines spent where, summed since the last reset (smm_debug_host_stats).
struct HostStats {
  std::mutex mu;
  double v[SMM_HOST_STAT_COUNT] = {};
} g_host_stats;

inline double wall_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// one call's share: host-side stages by the wall clock of the calling thread, device-side stages from the
// chunk's four events once its stream has been synchronised
struct CallStats {
  double v[SMM_HOST_STAT_COUNT] = {};
  double t_call = wall_ms();
  void chunk_done(HostPipe& pipe, int b) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pipe.ev[b][0], pipe.ev[b][1]) == hipSuccess) v[SMM_HOST_STAT_H2D_MS] += ms;
    if (hipEventElapsedTime(&ms, pipe.ev[b][1], pipe.ev[b][2]) == hipSuccess) v[SMM_HOST_STAT_KERNEL_MS] += ms;
    if (hipEventElapsedTime(&ms, pipe.ev[b][2], pipe.ev[b][3]) == hipSuccess) v[SMM_HOST_STAT_D2H_MS] += ms;
    (void)hipGetLastError();
    v[SMM_HOST_STAT_CHUNKS] += 1;
  }
  ~CallStats() {
    v[SMM_HOST_STAT_CALLS] = 1;
    v[SMM_HOST_STAT_TOTAL_MS] = wall_ms() - t_call;
    std::lock_guard<std::mutex> lock(g_host_stats.mu);
    for (int i = 0; i < SMM_HOST_STAT_COUNT; ++i) g_host_stats.v[i] += v[i];
    g_host_stats.v[SMM_HOST_STAT_THREADS] = (double)smm::staging_threads();
  }
};
struct StageTimer {   // adds the scope's wall time to one entry
  double& acc;
  double t0 = wall_ms();
  explicit StageTimer(double& a) : acc(a) {}
  ~StageTimer() { acc += wall_ms() - t0; }
};

bool is_pinned(const void* p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

}  // namespace

namespace {

// Every extern "C" entry that can reach an allocation runs its body through this: the header promises an int
// status, never an exception.  (fail() assigns a std::string and may itself run out of memory: then the status
// alone has to do.)
template <typename F>
int guarded(F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    try {
      return fail(SMM_ERR_ALLOC, "out of host memory");
    } catch (...) {
      return SMM_ERR_ALLOC;
    }
  } catch (const std::exception& e) {
    try {
      return fail(SMM_ERR_INTERNAL, std::string("unexpected failure: ") + e.what());
    } catch (...) {
      return SMM_ERR_INTERNAL;
    }
  } catch (...) {
    try {
      return fail(SMM_ERR_INTERNAL, "unexpected failure (unknown exception)");
    } catch (...) {
      return SMM_ERR_INTERNAL;
    }
  }
}

// apply flags the ABI defines; anything else (ABI v4 callers encoded kernel variants in bits 16..23) is refused
constexpr unsigned kApplyFlagMask = SMM_APPLY_MASKED | SMM_APPLY_NO_FILL | SMM_APPLY_SB_PACKED | SMM_APPLY_HOST_NO_PACK |
                                    SMM_APPLY_SB_Y_SB | SMM_APPLY_KERNEL_SELL | SMM_APPLY_KERNEL_TILE;
inline int check_flags(unsigned flags) {
  if (flags & ~kApplyFlagMask)
    return fail(SMM_ERR_INVALID, "unknown apply flag bits 0x" + [](unsigned v) {
             char buf[16];
             snprintf(buf, sizeof(buf), "%x", v);
             return std::string(buf);
           }(flags & ~kApplyFlagMask) + " (launch-shape knobs are smm_debug_set_tuning entries, not flags)");
  return SMM_OK;
}

}  // namespace

// ------------------------------------------------------------------ C ABI

extern "C" {

int smm_abi_version(void) { return SMM_ABI_VERSION; }
const char* smm_last_error(void) { return g_last_error.c_str(); }

int smm_device_count(int* count) {
  if (!count) return fail(SMM_ERR_INVALID, "null count");
  *count = 0;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(SMM_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  *count = n;
  return SMM_OK;
}
int smm_set_device(int device) {
  SMM_HIP(hipSetDevice(device));
  return SMM_OK;
}
int smm_get_device(int* device) {
  if (!device) return fail(SMM_ERR_INVALID, "null device");
  SMM_HIP(hipGetDevice(device));
  return SMM_OK;
}
int smm_device_name(int device, char* buf, size_t buflen) {
  if (!buf || buflen == 0) return fail(SMM_ERR_INVALID, "null buffer");
  hipDeviceProp_t p;
  SMM_HIP(hipGetDeviceProperties(&p, device));
  snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
  return SMM_OK;
}
int smm_mem_info(size_t* free_bytes, size_t* total_bytes) {
  size_t f = 0, t = 0;
  SMM_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return SMM_OK;
}
int smm_malloc(void** dptr, size_t bytes) {
  if (!dptr) return fail(SMM_ERR_INVALID, "null dptr");
  *dptr = nullptr;
  SMM_HIP(hipMalloc(dptr, bytes ? bytes : 1));
  return SMM_OK;
}
int smm_free(void* dptr) {
  if (dptr) SMM_HIP(hipFree(dptr));
  return SMM_OK;
}
int smm_host_alloc(void** hptr, size_t bytes) {
  if (!hptr) return fail(SMM_ERR_INVALID, "null hptr");
  *hptr = nullptr;
  SMM_HIP(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
  return SMM_OK;
}
int smm_host_free(void* hptr) {
  if (hptr) SMM_HIP(hipHostFree(hptr));
  return SMM_OK;
}
int smm_host_memcpy(void* dst_host, const void* src_host, size_t bytes) {
  if (bytes == 0) return SMM_OK;
  if (!dst_host || !src_host) return fail(SMM_ERR_INVALID, "null host pointer");
  return host_copy(dst_host, src_host, bytes);
}
int smm_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream) {
  if (bytes == 0) return SMM_OK;
  if (stream)
    SMM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  else
    SMM_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return SMM_OK;
}
int smm_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream) {
  if (bytes == 0) return SMM_OK;
  if (stream)
    SMM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  else
    SMM_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return SMM_OK;
}
int smm_memcpy_d2d(void* dst, const void* src, size_t bytes, void* stream) {
  if (bytes == 0) return SMM_OK;
  SMM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return SMM_OK;
}
int smm_memcpy2d_h2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                     void* stream) {
  if (width == 0 || height == 0) return SMM_OK;
  if (dpitch < width || spitch < width) return fail(SMM_ERR_INVALID, "pitch smaller than the row width");
  if (stream)
    SMM_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyHostToDevice, (hipStream_t)stream));
  else
    SMM_HIP(hipMemcpy2D(dst, dpitch, src, spitch, width, height, hipMemcpyHostToDevice));
  return SMM_OK;
}
int smm_memcpy2d_d2h(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                     void* stream) {
  if (width == 0 || height == 0) return SMM_OK;
  if (dpitch < width || spitch < width) return fail(SMM_ERR_INVALID, "pitch smaller than the row width");
  if (stream)
    SMM_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost, (hipStream_t)stream));
  else
    SMM_HIP(hipMemcpy2D(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToHost));
  return SMM_OK;
}
int smm_debug_set_grid_limit(int64_t max_blocks) {
  if (max_blocks < 0) return fail(SMM_ERR_INVALID, "negative grid limit");
  g_grid_limit.store(max_blocks == 0 || max_blocks > 0x7fffffffLL ? 0x7fffffffLL : max_blocks);
  return SMM_OK;
}

int smm_debug_set_tuning(int knob, int value, int* previous) {
  if (knob < 0 || knob >= SMM_TUNE_COUNT) return fail(SMM_ERR_INVALID, "unknown tuning knob " + std::to_string(knob));
  const int prev = g_tuning[knob].exchange(value);
  if (previous) *previous = prev;
  return SMM_OK;
}

int smm_set_host_threads(int n, int* previous) {
  if (n < 0) return fail(SMM_ERR_INVALID, "negative thread count");
  const int prev = smm::set_host_threads(n);
  if (previous) *previous = prev;
  return SMM_OK;
}

int smm_debug_host_stats(double* out, int n, int reset) {
  if (n < 0 || (n > 0 && !out)) return fail(SMM_ERR_INVALID, "bad stats buffer");
  std::lock_guard<std::mutex> lock(g_host_stats.mu);
  for (int i = 0; i < n; ++i) out[i] = i < SMM_HOST_STAT_COUNT ? g_host_stats.v[i] : 0.0;
  if (reset)
    for (double& v : g_host_stats.v) v = 0.0;
  return SMM_OK;
}

int smm_debug_staging_faults(int no_threads, int64_t throw_in_task) {
  smm::debug_pool_faults(no_threads != 0, throw_in_task < 0 ? -1 : throw_in_task);
  return SMM_OK;
}

int smm_debug_fail_at_chunk(int64_t chunk) {
  g_fail_at_chunk.store(chunk < 0 ? -1 : chunk, std::memory_order_relaxed);
  return SMM_OK;
}
int smm_memset(void* dst, int value, size_t bytes, void* stream) {
  if (bytes == 0) return SMM_OK;
  SMM_HIP(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
  return SMM_OK;
}
int smm_stream_create(void** stream) {
  if (!stream) return fail(SMM_ERR_INVALID, "null stream");
  hipStream_t s;
  SMM_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void*)s;
  return SMM_OK;
}
int smm_stream_destroy(void* stream) {
  if (stream) SMM_HIP(hipStreamDestroy((hipStream_t)stream));
  return SMM_OK;
}
int smm_stream_sync(void* stream) {
  SMM_HIP(hipStreamSynchronize((hipStream_t)stream));
  return SMM_OK;
}
int smm_device_sync(void) {
  SMM_HIP(hipDeviceSynchronize());
  return SMM_OK;
}
int smm_event_create(void** event) {
  if (!event) return fail(SMM_ERR_INVALID, "null event");
  hipEvent_t e;
  SMM_HIP(hipEventCreate(&e));
  *event = (void*)e;
  return SMM_OK;
}
int smm_event_destroy(void* event) {
  if (event) SMM_HIP(hipEventDestroy((hipEvent_t)event));
  return SMM_OK;
}
int smm_event_record(void* event, void* stream) {
  SMM_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return SMM_OK;
}
int smm_event_sync(void* event) {
  SMM_HIP(hipEventSynchronize((hipEvent_t)event));
  return SMM_OK;
}
int smm_stream_wait_event(void* stream, void* event) {
  SMM_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
  return SMM_OK;
}
int smm_event_elapsed_ms(void* start, void* stop, float* ms) {
  if (!ms) return fail(SMM_ERR_INVALID, "null ms");
  SMM_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return SMM_OK;
}

int smm_fill_random(void* dst, int dtype, int64_t n, uint64_t seed, double mean, double sigma,
                    void* stream) {
  if (n <= 0) return SMM_OK;
  if (!dst) return fail(SMM_ERR_INVALID, "null destination");
  const unsigned blocks = (unsigned)std::min<int64_t>((n + 255) / 256, 256 * 32);
  if (dtype == SMM_F64)
    hipLaunchKernelGGL(smm_fill_random_kernel<double>, dim3(blocks), dim3(256), 0,
                       (hipStream_t)stream, (double*)dst, n, seed, mean, sigma);
  else if (dtype == SMM_F32)
    hipLaunchKernelGGL(smm_fill_random_kernel<float>, dim3(blocks), dim3(256), 0,
                       (hipStream_t)stream, (float*)dst, n, seed, mean, sigma);
  else
    return fail(SMM_ERR_UNSUPPORTED, "dtype must be SMM_F32 or SMM_F64");
  SMM_HIP(hipGetLastError());
  return SMM_OK;
}

// ---- operators

// Shared by the two constructors: `fill_csr` builds op->csr (false + err on invalid input).
extern "C++" {
template <typename F>
static int create_operator(int device, smm_operator_t* out, F fill_csr, unsigned options = 0u) {
  if (!out) return fail(SMM_ERR_INVALID, "null out handle");
  *out = nullptr;
  if (options & ~(unsigned)SMM_CREATE_PRUNE_ZEROS) return fail(SMM_ERR_INVALID, "unknown create option bits");
  int ndev = 0;
  {
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
      (void)hipGetLastError();
      return fail(SMM_ERR_NO_DEVICE,
                  "no HIP device: libsmmregrid_hip has no CPU fallback (hipGetDeviceCount: " +
                      std::string(hipGetErrorString(e)) + ")");
    }
  }
  if (device < 0 || device >= ndev)
    return fail(SMM_ERR_NO_DEVICE, "device ordinal " + std::to_string(device) + " out of range");

  smm_operator* op = new (std::nothrow) smm_operator();
  if (!op) return fail(SMM_ERR_ALLOC, "out of host memory");
  op->device = device;
  try {
    std::string err;
    if (!fill_csr(op->csr, err)) {
      delete op;
      return fail(SMM_ERR_INVALID, err);
    }
    if (options & SMM_CREATE_PRUNE_ZEROS) op->pruned_links = smm::prune_zero_links(op->csr);
    const smm::HostCsr& kc = op->csr;
    smm::HostSell sell;
    smm::build_sell(kc, sell);

    DeviceGuard guard(device);
    if (!guard.ok) {
      delete op;
      return fail(SMM_ERR_HIP, "cannot select device " + std::to_string(device));
    }
    op->n_slices = sell.n_slices;
    op->n_slots = sell.n_slots;
    int rc = SMM_OK;
    if ((rc = upload(&op->d_slice_off, sell.slice_off)) || (rc = upload(&op->d_col, sell.col)) ||
        (rc = upload(&op->d_val, sell.val)) || (rc = upload(&op->d_rowlen, sell.rowlen))) {
      release(op);
      return rc;
    }
    op->sell_shape.n_slices = sell.n_slices;
    op->sell_shape.n_slots = sell.n_slots;
    op->sell_shape.slice_off = std::move(sell.slice_off);
    op->sell_shape.rowlen = std::move(sell.rowlen);
    // own block shape: 256 rows for rows of <= 16 links; else one slice, or the largest part of a
    // slice whose footprint fits the LDS budget and is used well enough (plan valid and preferred)
    op->native = kc.max_row_nnz > 16 ? 1 : 0;
    if ((rc = ensure_plan(op, op->native))) {
      release(op);
      return rc;
    }
    if (op->native == 1) {
      // rows beyond 48 links: start at the shape whose lane groups can keep the whole row in
      // registers (split rows) -- streaming the links from L2 is ~2x slower; else from one slice
      int w_first = 1;
      while (w_first < kNumShapes - 1 && kc.max_row_nnz > 48ll << (w_first - 1)) ++w_first;
      bool found = false;
      for (int pass = 0; pass < 2 && !found; ++pass) {
        for (int w = pass == 0 ? w_first : 1; w < (pass == 0 ? kNumShapes : w_first) && !found; ++w) {
          if ((rc = ensure_plan(op, w))) {
            release(op);
            return rc;
          }
          if (op->plan[w].valid && op->plan[w].preferred) {
            op->native = w;
            found = true;
          }
        }
      }
      for (int w = 1; w < kNumShapes; ++w) {   // plans tried on the way are rebuilt on demand
        if (w == op->native) continue;
        smm_operator::TilePlan& pl = op->plan[w];
        (void)hipFree(pl.d_blk_chunk_off);
        (void)hipFree(pl.d_chunk_src);
        (void)hipFree(pl.d_lcol);
        (void)hipFree(pl.d_blk_direct);
        pl = smm_operator::TilePlan();
      }
    }
    if ((rc = refresh_desc(op))) {
      release(op);
      return rc;
    }
  } catch (const std::bad_alloc&) {
    release(op);
    return fail(SMM_ERR_ALLOC, "out of host memory while building the operator");
  } catch (const std::exception& e) {   // nothing may cross the extern "C" boundary
    release(op);
    return fail(SMM_ERR_INTERNAL, std::string("operator build failed: ") + e.what());
  } catch (...) {
    release(op);
    return fail(SMM_ERR_INTERNAL, "operator build failed");
  }
  *out = op;
  return SMM_OK;
}
}  // extern "C++"

int smm_operator_create(int64_t n_src, int64_t n_dst, int64_t nnz, const int32_t* src_addr_1based,
                        const int32_t* dst_addr_1based, const double* w, int device,
                        smm_operator_t* out) {
  return create_operator(device, out, [&](smm::HostCsr& csr, std::string& err) {
    return smm::build_csr(n_src, n_dst, nnz, src_addr_1based, dst_addr_1based, w, csr, err);
  });
}

int smm_operator_create_opt(int64_t n_src, int64_t n_dst, int64_t nnz, const int32_t* src_addr_1based,
                            const int32_t* dst_addr_1based, const double* w, unsigned options, int device,
                            smm_operator_t* out) {
  return create_operator(device, out, [&](smm::HostCsr& csr, std::string& err) {
    return smm::build_csr(n_src, n_dst, nnz, src_addr_1based, dst_addr_1based, w, csr, err);
  }, options);
}

int smm_operator_create_csr(int64_t n_src, int64_t n_dst, const int64_t* rowptr, const int32_t* col,
                            const double* val, int device, smm_operator_t* out) {
  return create_operator(device, out, [&](smm::HostCsr& csr, std::string& err) {
    return smm::adopt_csr(n_src, n_dst, rowptr, col, val, csr, err);
  });
}

int smm_operator_destroy(smm_operator_t op) {
  if (!op) return SMM_OK;
  if (op->group_refs.load() > 0)
    return fail(SMM_ERR_INVALID, "operator still belongs to a group: destroy the group first");
  DeviceGuard guard(op->device);
  release(op);
  return SMM_OK;
}

int smm_operator_info(smm_operator_t op, int64_t* n_src, int64_t* n_dst, int64_t* nnz,
                      int64_t* n_used_src, int64_t* max_row_nnz) {
  if (!op) return fail(SMM_ERR_INVALID, "null operator");
  if (n_src) *n_src = op->csr.n_src;
  if (n_dst) *n_dst = op->csr.n_dst;
  if (nnz) *nnz = op->csr.nnz;
  if (n_used_src) *n_used_src = op->csr.n_used_src;
  if (max_row_nnz) *max_row_nnz = op->csr.max_row_nnz;
  return SMM_OK;
}

static int smm_operator_export_csr_impl(smm_operator_t op, int64_t* rowptr, int32_t* col, double* val) {
  if (!op) return fail(SMM_ERR_INVALID, "null operator");
  if (rowptr) memcpy(rowptr, op->csr.rowptr.data(), op->csr.rowptr.size() * sizeof(int64_t));
  if (col && op->csr.nnz) memcpy(col, op->csr.col.data(), (size_t)op->csr.nnz * sizeof(int32_t));
  if (val && op->csr.nnz) memcpy(val, op->csr.val.data(), (size_t)op->csr.nnz * sizeof(double));
  return SMM_OK;
}

static int smm_operator_set_epilogue_impl(smm_operator_t op, const int32_t* dst_imask, const double* dst_frac) {
  if (!op) return fail(SMM_ERR_INVALID, "null operator");
  // a group's level descriptors hold this operator's imask / frac device pointers
  if (op->group_refs.load() > 0)
    return fail(SMM_ERR_INVALID,
                "operator belongs to a group: set the epilogue vectors before smm_group_create");
  DeviceGuard guard(op->device);
  if (!guard.ok) return fail(SMM_ERR_HIP, "cannot select the operator's device");
  const size_t n = (size_t)op->csr.n_dst;
  // upload the new vectors first: on failure the operator keeps its old state untouched
  uint8_t* new_imask = nullptr;
  double* new_frac = nullptr;
  auto drop_new = [&]() {
    (void)hipFree(new_imask);
    (void)hipFree(new_frac);
  };
  if (dst_imask) {
    std::vector<uint8_t> m(n);
    for (size_t i = 0; i < n; ++i) m[i] = dst_imask[i] != 0;  // .astype(bool), regrid.py:557
    int rc = upload(&new_imask, m);
    if (rc) {
      drop_new();
      return rc;
    }
  }
  if (dst_frac) {
    std::vector<double> f(dst_frac, dst_frac + n);
    int rc = upload(&new_frac, f);
    if (rc) {
      drop_new();
      return rc;
    }
  }
  uint8_t* old_imask = op->d_imask;
  double* old_frac = op->d_frac;
  op->d_imask = new_imask;
  op->d_frac = new_frac;
  int rc = refresh_desc(op);
  if (rc) {  // the device descriptor still names the old vectors: keep them
    op->d_imask = old_imask;
    op->d_frac = old_frac;
    drop_new();
    return rc;
  }
  (void)hipFree(old_imask);
  (void)hipFree(old_frac);
  return SMM_OK;
}

static int smm_operator_plan_info_impl(smm_operator_t op, int* kernel_kind, int64_t* lds_bytes,
                           int64_t* staged_src_elems) {
  if (!op) return fail(SMM_ERR_INVALID, "null operator");
  const smm_operator::TilePlan& pl = op->plan[op->native_plan()];
  if (kernel_kind)
    *kernel_kind = (pl.valid ? 1 : 0) | (pl.preferred ? 2 : 0) | (shape_rows(op->native_plan()) << 8);
  if (lds_bytes) *lds_bytes = pl.valid ? pl.max_chunks * kChunkElems * 8 : 0;
  if (staged_src_elems) *staged_src_elems = pl.valid ? pl.total_lines * 16 : 0;   // whole 128-B lines of f64
  return SMM_OK;
}

static int smm_apply_impl(smm_operator_t op, const void* x, int x_dtype, int64_t ldx, void* y, int y_dtype,
              int64_t ldy, int64_t n_batch, double remap_area_min, unsigned flags, void* stream) {
  if (int frc = check_flags(flags)) return frc;
  if (!op) return fail(SMM_ERR_INVALID, "null operator");
  if (n_batch > 0 && (ldx < op->csr.n_src || ldy < op->csr.n_dst))
    return fail(SMM_ERR_INVALID, "ldx/ldy smaller than the grid size");
  if ((flags & SMM_APPLY_MASKED) && !op->d_imask)
    return fail(SMM_ERR_INVALID, "masked apply requested but the operator has no dst_imask");
  if (remap_area_min > 0.0 && !op->d_frac)
    return fail(SMM_ERR_INVALID, "remap_area_min > 0 requested but the operator has no dst_frac");
  DeviceGuard guard(op->device);
  if (!guard.ok) return fail(SMM_ERR_HIP, "cannot select the operator's device");
  const int pw = op->native_plan();
  const smm_operator::TilePlan& pl = op->plan[pw];
  return run_apply(op->d_desc, nullptr, nullptr, op->csr.n_src, op->csr.n_dst, pw, pl.valid,
                   pl.preferred, (pl.reuse ? 1 : 0), pl.max_chunks, op->csr.max_row_nnz, x, x_dtype, ldx, 0, 0, y, y_dtype, ldy,
                   0, 0, n_batch, 1, 1, remap_area_min, flags, (hipStream_t)stream);
}

static int smm_operator_prepare_sb_impl(smm_operator_t op) {
  if (!op) return fail(SMM_ERR_INVALID, "null operator");
  DeviceGuard guard(op->device);
  if (!guard.ok) return fail(SMM_ERR_HIP, "cannot select the operator's device");
  return ensure_sb(op);
}

static int smm_operator_used_sources_impl(smm_operator_t op, int32_t* used) {
  if (!op) return fail(SMM_ERR_INVALID, "null operator");
  if (!used && op->csr.n_used_src > 0) return fail(SMM_ERR_INVALID, "null output");
  std::vector<uint8_t> seen((size_t)std::max<int64_t>(op->csr.n_src, 1), 0);
  for (int32_t s : op->csr.col) seen[(size_t)s] = 1;
  int64_t k = 0;
  for (int64_t s = 0; s < op->csr.n_src; ++s)
    if (seen[(size_t)s]) used[k++] = (int32_t)s;
  return SMM_OK;
}

static int smm_apply_sb_impl(smm_operator_t op, const void* x, int x_dtype, int64_t ldx, void* y, int y_dtype,
                 int64_t ldy, int64_t n_batch, double remap_area_min, unsigned flags, void* stream) {
  if (int frc = check_flags(flags)) return frc;
  if (!op) return fail(SMM_ERR_INVALID, "null operator");
  if (n_batch < 0) return fail(SMM_ERR_INVALID, "negative batch size");
  if (n_batch == 0 || op->csr.n_dst == 0) return SMM_OK;
  if (!x || !y) return fail(SMM_ERR_INVALID, "null field pointer");
  if ((x_dtype != SMM_F32 && x_dtype != SMM_F64) || (y_dtype != SMM_F32 && y_dtype != SMM_F64))
    return fail(SMM_ERR_UNSUPPORTED, "field dtype must be SMM_F32 or SMM_F64");
  if (ldx < n_batch || ldy < ((flags & SMM_APPLY_SB_Y_SB) ? n_batch : op->csr.n_dst))
    return fail(SMM_ERR_INVALID, "ldx smaller than the batch or ldy smaller than a row of Y");
  if (!(remap_area_min >= 0.0 && remap_area_min <= 1.0))
    return fail(SMM_ERR_INVALID, "remap_area_min must be within [0, 1]");  // regrid.py:124-125
  if ((flags & SMM_APPLY_MASKED) && !op->d_imask)
    return fail(SMM_ERR_INVALID, "masked apply requested but the operator has no dst_imask");
  if (remap_area_min > 0.0 && !op->d_frac)
    return fail(SMM_ERR_INVALID, "remap_area_min > 0 requested but the operator has no dst_frac");
  const size_t xsz = x_dtype == SMM_F64 ? 8 : 4, ysz = y_dtype == SMM_F64 ? 8 : 4;
  if ((uintptr_t)x % xsz || (uintptr_t)y % ysz) return fail(SMM_ERR_INVALID, "field pointer is not element aligned");
  DeviceGuard guard(op->device);
  if (!guard.ok) return fail(SMM_ERR_HIP, "cannot select the operator's device");
  int rc = ensure_sb(op);   // first call uploads the CSR (smm_operator_prepare_sb does it ahead of time)
  if (rc) return rc;
  SbArgs a{};
  a.rowptr = op->d_csr_rowptr;
  a.col = (flags & SMM_APPLY_SB_PACKED) ? op->d_csr_colp : op->d_csr_col;
  a.val = op->d_csr_val;
  a.imask = op->d_imask;
  a.frac = op->d_frac;
  a.x = x;
  a.y = y;
  a.ldx = ldx;
  a.ldy = ldy;
  a.n_batch = n_batch;
  a.n_dst = op->csr.n_dst;
  a.area_min = remap_area_min;
  a.masked = (flags & SMM_APPLY_MASKED) ? 1 : 0;
  const bool fill = !(flags & SMM_APPLY_NO_FILL);
  hipStream_t s = (hipStream_t)stream;
  // grid = destination tiles x batch tiles of 128 entries: beyond the limit the batch is cut into runs of
  // whole batch tiles, launched one after the other
  const int64_t n_dtiles = (a.n_dst + (ysz == 8 ? 16 : 32) - 1) / (ysz == 8 ? 16 : 32);
  const int64_t limit = grid_limit();
  if (n_dtiles > limit)
    return fail(SMM_ERR_INVALID, "one batch tile alone needs a launch grid beyond " + std::to_string(limit) + " workgroups");
  const int64_t part = std::max<int64_t>(1, limit / n_dtiles) * 128;
  for (int64_t b0 = 0; b0 < n_batch; b0 += part) {
    a.x = (const char*)x + b0 * (int64_t)xsz;
    a.y = (char*)y + ((flags & SMM_APPLY_SB_Y_SB) ? b0 : b0 * ldy) * (int64_t)ysz;
    a.n_batch = std::min(part, n_batch - b0);
    rc = x_dtype == SMM_F64
             ? (y_dtype == SMM_F64 ? launch_sb<double, double>(a, fill, flags, s) : launch_sb<double, float>(a, fill, flags, s))
             : (y_dtype == SMM_F64 ? launch_sb<float, double>(a, fill, flags, s) : launch_sb<float, float>(a, fill, flags, s));
    if (rc) return rc;
  }
  return SMM_OK;
}

// ---- host-buffer path: chunked, double-buffered H2D -> kernel -> D2H pipeline

static int smm_apply_host_impl(smm_operator_t op, const void* x_host, int x_dtype, int64_t ldx, void* y_host,
                   int y_dtype, int64_t ldy, int64_t n_batch, double remap_area_min, unsigned flags,
                   int64_t chunk_rows) {
  if (int frc = check_flags(flags)) return frc;
  if (!op) return fail(SMM_ERR_INVALID, "null operator");
  if (n_batch < 0) return fail(SMM_ERR_INVALID, "negative batch size");
  if (n_batch == 0 || op->csr.n_dst == 0) return SMM_OK;
  if (!x_host || !y_host) return fail(SMM_ERR_INVALID, "null field pointer");
  if ((x_dtype != SMM_F32 && x_dtype != SMM_F64) || (y_dtype != SMM_F32 && y_dtype != SMM_F64))
    return fail(SMM_ERR_UNSUPPORTED, "field dtype must be SMM_F32 or SMM_F64");
  const int64_t S = op->csr.n_src, D = op->csr.n_dst;
  if (ldx < S || ldy < D) return fail(SMM_ERR_INVALID, "ldx/ldy smaller than the grid size");
  DeviceGuard guard(op->device);
  if (!guard.ok) return fail(SMM_ERR_HIP, "cannot select the operator's device");

  const size_t xsz = x_dtype == SMM_F64 ? 8 : 4, ysz = y_dtype == SMM_F64 ? 8 : 4;
  const size_t xrow = (size_t)ldx * xsz, yrow = (size_t)ldy * ysz;   // host row pitches
  // device rows start on 128-B lines: the tile plan stages whole lines of a row, and a row that
  // starts mid-line makes every staged run of chunks straddle one line more (config 3's 1442x1021
  // source: +17 % fetched bytes when its rows are packed back to back)
  const size_t xrow_d = (((size_t)S * xsz + 127) / 128) * 128;
  const int64_t ldx_d = (int64_t)(xrow_d / xsz);
  // Operators that use at most four fifths of their source cells (round 6: half before; bilinear / nearest downsampling: config 2
  // uses a quarter) do not ship the whole field over PCIe: the staging copy packs the used cells of a
  // chunk batch-fastest (host_pack) and the chunk runs through the batch-fastest kernel.  Same bits.
  const int64_t U = op->csr.n_used_src;
  const bool may_pack = !(flags & (SMM_APPLY_HOST_NO_PACK | SMM_APPLY_KERNEL_SELL | SMM_APPLY_KERNEL_TILE)) &&
                        U > 0 && U * 5 <= S * 4;
  // Chunk size from the X AND Y bytes of a row (an operator with few used cells and a large target
  // is bound by its Y staging), clamped to a quarter of the free device memory: smm_internal.h
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    free_b = 0;
  }
  const smm::HostChunk hc = smm::host_chunk_units(n_batch, xrow_d, (size_t)D * ysz, may_pack ? (size_t)U * xsz : 0,
                                                  8, 128, chunk_rows, free_b);   // packing pays from 8 rows on (24 rows: 4.2 -> 1.9 ms)
  const bool pack = hc.pack;
  chunk_rows = hc.units;
  if (pack) {
    int prc = ensure_sb(op);
    if (prc) return prc;
  }
  const bool x_pinned = is_pinned(x_host), y_pinned = is_pinned(y_host);
  // a pinned source with the device pitch can be DMA'd row-block-wise without staging
  const bool x_direct = x_pinned && !pack, y_direct = y_pinned;

  std::lock_guard<std::mutex> pipe_lock(op->pipe_mu);
  HostPipe& pipe = op->pipe;
  const size_t x_chunk_d = pack ? (size_t)chunk_rows * U * xsz : (size_t)chunk_rows * xrow_d;
  SMM_HIP(pipe.ensure(x_chunk_d, (size_t)chunk_rows * D * ysz,
                      x_direct ? 0 : (pack ? x_chunk_d : (size_t)chunk_rows * S * xsz),
                      y_direct ? 0 : (size_t)chunk_rows * D * ysz));

  const int64_t n_chunks = (n_batch + chunk_rows - 1) / chunk_rows;
  CallStats st;
  auto drain = [&](int64_t c) -> int {  // results of chunk c: wait, then pinned -> user rows
    const int b = (int)(c & 1);
    {
      StageTimer t(st.v[SMM_HOST_STAT_WAIT_MS]);
      SMM_HIP(hipStreamSynchronize(pipe.stream[b]));
    }
    st.chunk_done(pipe, b);
    if (!y_direct) {
      StageTimer t(st.v[SMM_HOST_STAT_COPY_OUT_MS]);
      const int64_t r0 = c * chunk_rows, rows = std::min(chunk_rows, n_batch - r0);
      if ((int64_t)ldy == D) {
        int rc = host_copy((char*)y_host + (size_t)r0 * yrow, pipe.hy[b], (size_t)rows * D * ysz);
        if (rc) return rc;
      } else {
        for (int64_t r = 0; r < rows; ++r)
          memcpy((char*)y_host + (size_t)(r0 + r) * yrow, (char*)pipe.hy[b] + (size_t)r * D * ysz,
                 (size_t)D * ysz);
      }
    }
    return SMM_OK;
  };

  // The chunk loop runs inside a lambda so that any failure can first wait for the copies still
  // in flight into the caller's buffers (chunk c-1's D2H) before the error is returned.
  const int64_t fail_at = test_fail_chunk();
  auto pipeline = [&]() -> int {
  for (int64_t c = 0; c < n_chunks; ++c) {
    const int b = (int)(c & 1);
    const int64_t r0 = c * chunk_rows, rows = std::min(chunk_rows, n_batch - r0);
    if (c >= 2) {
      int rc = drain(c - 2);  // buffer b is free again once chunk c-2 has been delivered
      if (rc) return rc;
    }
    if (c == fail_at) return fail(SMM_ERR_HIP, "injected failure (smm_debug_fail_at_chunk)");
    const char* xsrc = (const char*)x_host + (size_t)r0 * xrow;
    if (pack) {
      {
        StageTimer t(st.v[SMM_HOST_STAT_STAGE_IN_MS]);
        int rc = host_pack(pipe.hx[b], xsrc, xsz, ldx, op->h_used, rows);
        if (rc) return rc;
      }
      SMM_HIP(hipEventRecord(pipe.ev[b][0], pipe.stream[b]));
      SMM_HIP(hipMemcpyAsync(pipe.dx[b], pipe.hx[b], (size_t)U * rows * xsz, hipMemcpyHostToDevice, pipe.stream[b]));
    } else if (!x_direct) {
      {
        StageTimer t(st.v[SMM_HOST_STAT_STAGE_IN_MS]);
        if (ldx == S) {
          int rc = host_copy(pipe.hx[b], xsrc, (size_t)rows * S * xsz);
          if (rc) return rc;
        } else {
          for (int64_t r = 0; r < rows; ++r)
            memcpy((char*)pipe.hx[b] + (size_t)r * S * xsz, xsrc + (size_t)r * xrow, (size_t)S * xsz);
        }
      }
      SMM_HIP(hipEventRecord(pipe.ev[b][0], pipe.stream[b]));
      SMM_HIP(hipMemcpy2DAsync(pipe.dx[b], xrow_d, pipe.hx[b], (size_t)S * xsz, (size_t)S * xsz,
                               (size_t)rows, hipMemcpyHostToDevice, pipe.stream[b]));
    } else {
      SMM_HIP(hipEventRecord(pipe.ev[b][0], pipe.stream[b]));
      SMM_HIP(hipMemcpy2DAsync(pipe.dx[b], xrow_d, xsrc, xrow, (size_t)S * xsz, (size_t)rows,
                               hipMemcpyHostToDevice, pipe.stream[b]));
    }
    SMM_HIP(hipEventRecord(pipe.ev[b][1], pipe.stream[b]));
    const int pw = op->native_plan();
    const smm_operator::TilePlan& pl = op->plan[pw];
    int rc = SMM_OK;
    if (pack)
      rc = smm_apply_sb_impl(op, pipe.dx[b], x_dtype, rows, pipe.dy[b], y_dtype, D, rows, remap_area_min,
                         (flags & (SMM_APPLY_MASKED | SMM_APPLY_NO_FILL)) | SMM_APPLY_SB_PACKED, pipe.stream[b]);
    else
      rc = run_apply(op->d_desc, nullptr, nullptr, S, op->csr.n_dst, pw, pl.valid, pl.preferred, (pl.reuse ? 1 : 0),
                     pl.max_chunks, op->csr.max_row_nnz, pipe.dx[b], x_dtype,
                     ldx_d, 0, 0, pipe.dy[b], y_dtype, D, 0, 0, rows, 1, 1, remap_area_min, flags,
                     pipe.stream[b]);
    if (rc) return rc;
    SMM_HIP(hipEventRecord(pipe.ev[b][2], pipe.stream[b]));
    if (!y_direct) {
      SMM_HIP(hipMemcpyAsync(pipe.hy[b], pipe.dy[b], (size_t)rows * D * ysz, hipMemcpyDeviceToHost,
                             pipe.stream[b]));
    } else {
      SMM_HIP(hipMemcpy2DAsync((char*)y_host + (size_t)r0 * yrow, yrow, pipe.dy[b], (size_t)D * ysz,
                               (size_t)D * ysz, (size_t)rows, hipMemcpyDeviceToHost, pipe.stream[b]));
    }
    SMM_HIP(hipEventRecord(pipe.ev[b][3], pipe.stream[b]));
  }
  for (int64_t c = std::max<int64_t>(0, n_chunks - 2); c < n_chunks; ++c) {
    int rc = drain(c);
    if (rc) return rc;
  }
  return SMM_OK;
  };
  const int prc = pipeline();
  if (prc) pipe.quiesce();   // keeps the thread's error message of the first failure
  return prc;
}

static int smm_operator_mask_apply_impl(smm_operator_t op, const int32_t* src_imask, int32_t* dst_imask) {
  if (!op || !src_imask || !dst_imask) return fail(SMM_ERR_INVALID, "null argument");
  DeviceGuard guard(op->device);
  if (!guard.ok) return fail(SMM_ERR_HIP, "cannot select the operator's device");
  const int64_t S = op->csr.n_src, D = op->csr.n_dst;
  if (D == 0) return SMM_OK;
  // int32 mask promoted to f64 for the product (weights.py:50)
  std::vector<double> xs((size_t)std::max<int64_t>(S, 1));
  for (int64_t i = 0; i < S; ++i) xs[(size_t)i] = (double)src_imask[i];
  double *dx = nullptr, *dy = nullptr;
  int32_t* dm = nullptr;
  int rc = SMM_OK;
  auto cleanup = [&]() {
    (void)hipFree(dx);
    (void)hipFree(dy);
    (void)hipFree(dm);
  };
  if (hipMalloc((void**)&dx, xs.size() * 8) != hipSuccess ||
      hipMalloc((void**)&dy, (size_t)D * 8) != hipSuccess ||
      hipMalloc((void**)&dm, (size_t)D * 4) != hipSuccess) {
    cleanup();
    (void)hipGetLastError();
    return fail(SMM_ERR_HIP, "hipMalloc failed in smm_operator_mask_apply");
  }
  if (hipMemcpy(dx, xs.data(), xs.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
    cleanup();
    return fail(SMM_ERR_HIP, "hipMemcpy failed in smm_operator_mask_apply");
  }
  rc = run_apply(op->d_desc, nullptr, nullptr, S, op->csr.n_dst, 0, false, false, false, 0, op->csr.max_row_nnz, dx, SMM_F64,
                 std::max<int64_t>(S, 1), 0, 0, dy, SMM_F64, D, 0, 0, 1, 1, 1, 0.0,
                 SMM_APPLY_NO_FILL, nullptr);
  if (rc == SMM_OK) {
    const int threads = 256;
    hipLaunchKernelGGL(smm_mask_threshold_kernel, dim3((unsigned)((D + threads - 1) / threads)),
                       dim3(threads), 0, nullptr, dy, dm, D);
    if (hipGetLastError() != hipSuccess ||
        hipMemcpy(dst_imask, dm, (size_t)D * 4, hipMemcpyDeviceToHost) != hipSuccess) {
      rc = fail(SMM_ERR_HIP, "mask threshold kernel / copy failed");
    }
  }
  cleanup();
  return rc;
}

// ---- groups

static int smm_group_create_impl(const smm_operator_t* ops, int n_ops, smm_group_t* out) {
  if (!out) return fail(SMM_ERR_INVALID, "null out handle");
  *out = nullptr;
  if (!ops || n_ops <= 0) return fail(SMM_ERR_INVALID, "a group needs at least one operator");
  for (int i = 0; i < n_ops; ++i) {
    if (!ops[i]) return fail(SMM_ERR_INVALID, "null operator in group");
    if (ops[i]->device != ops[0]->device || ops[i]->csr.n_src != ops[0]->csr.n_src ||
        ops[i]->csr.n_dst != ops[0]->csr.n_dst)
      return fail(SMM_ERR_INVALID, "group members must share device and grid sizes");
  }
  std::unique_ptr<smm_group> owner(new (std::nothrow) smm_group());   // freed on every early return and on a throw
  if (!owner) return fail(SMM_ERR_ALLOC, "out of host memory");
  smm_group* g = owner.get();
  g->device = ops[0]->device;
  g->ops.assign(ops, ops + n_ops);
  g->tile_valid = true;
  // one launch covers all levels: one plan shape for all members (single-wave blocks as soon as
  // any level has rows longer than 16 links), and the links' majority decides tile vs SELL
  for (int i = 0; i < n_ops; ++i) g->tile_which = std::max(g->tile_which, ops[i]->native_plan());
  DeviceGuard guard(g->device);
  if (!guard.ok) return fail(SMM_ERR_HIP, "cannot select device");
  int64_t nnz_all = 0, nnz_pref = 0;
  std::vector<LevelDesc> descs((size_t)n_ops);
  for (int i = 0; i < n_ops; ++i) {
    int prc = ensure_plan(ops[i], g->tile_which);
    if (prc) return prc;
    const smm_operator::TilePlan& pl = ops[i]->plan[g->tile_which];
    descs[(size_t)i] = ops[i]->desc(g->tile_which);
    g->tile_valid = g->tile_valid && pl.valid;
    nnz_all += ops[i]->csr.nnz;
    if (pl.preferred) nnz_pref += ops[i]->csr.nnz;
    g->tile_reuse = g->tile_reuse || pl.reuse;
    g->tile_max_chunks = std::max(g->tile_max_chunks, pl.max_chunks);
    g->max_row_nnz = std::max(g->max_row_nnz, ops[i]->csr.max_row_nnz);
  }
  g->tile_preferred = 2 * nnz_pref >= nnz_all;
  int rc = upload(&g->d_descs, descs);
  if (rc) return rc;
  for (int i = 0; i < n_ops; ++i) ops[i]->group_refs.fetch_add(1);
  *out = owner.release();
  return SMM_OK;
}

int smm_group_plan_info(smm_group_t g, int* kernel_kind, int* slices_per_block) {
  if (!g) return fail(SMM_ERR_INVALID, "null group");
  if (kernel_kind) *kernel_kind = (g->tile_valid ? 1 : 0) | (g->tile_preferred ? 2 : 0);
  if (slices_per_block) *slices_per_block = g->tile_which ? 1 : kWavesPerBlock;   // 1 also for parts of a slice
  return SMM_OK;
}

int smm_group_destroy(smm_group_t g) {
  if (!g) return SMM_OK;
  DeviceGuard guard(g->device);
  for (auto& kv : g->cfg_cache) (void)hipFree(kv.second);
  (void)hipFree(g->d_descs);
  for (smm_operator_t op : g->ops) op->group_refs.fetch_sub(1);
  delete g;
  return SMM_OK;
}

extern "C++" {
// Validates (level_index, masked_levels) against the group and returns the device copy of that
// configuration, uploading it on first sight.  Entries live until smm_group_destroy.
static int group_level_cfg(smm_group_t g, int64_t n_lev, const int32_t* level_index,
                           const uint8_t* masked_levels, double remap_area_min, unsigned flags,
                           const int32_t** d_map, const uint8_t** d_masked) {
  *d_map = nullptr;
  *d_masked = nullptr;
  if (!g) return fail(SMM_ERR_INVALID, "null group");
  if (n_lev < 0) return fail(SMM_ERR_INVALID, "negative level count");
  if (n_lev > 0 && !level_index) return fail(SMM_ERR_INVALID, "null level_index");
  const int n_ops = (int)g->ops.size();
  for (int64_t l = 0; l < n_lev; ++l) {
    const int32_t w = level_index[l];
    if (w < 0 || w >= n_ops)
      return fail(SMM_ERR_INVALID, "level_index[" + std::to_string(l) + "]=" + std::to_string(w) +
                                       " outside the group");
    const smm_operator* op = g->ops[(size_t)w];
    const bool m = (flags & SMM_APPLY_MASKED) && (!masked_levels || masked_levels[w]);
    if (m && !op->d_imask)
      return fail(SMM_ERR_INVALID, "masked apply requested but a level has no dst_imask");
    if (remap_area_min > 0.0 && !op->d_frac)
      return fail(SMM_ERR_INVALID, "remap_area_min > 0 requested but a level has no dst_frac");
  }
  if (n_lev == 0) return SMM_OK;

  // key: level count, then the map, then (if given) one masked flag per member -- unambiguous
  std::string key((const char*)&n_lev, sizeof(n_lev));
  key.append((const char*)level_index, (size_t)n_lev * sizeof(int32_t));
  key.push_back(masked_levels ? 1 : 0);
  if (masked_levels) key.append((const char*)masked_levels, (size_t)n_ops);
  void* d_cfg = nullptr;
  const size_t map_bytes = ((size_t)n_lev * 4 + 15) & ~(size_t)15;
  {
    std::lock_guard<std::mutex> lock(g->mu);
    auto it = g->cfg_cache.find(key);
    if (it != g->cfg_cache.end()) {
      d_cfg = it->second;
    } else {
      // first sight of this configuration (smm_group_prepare does this ahead of time): one small
      // allocation + blocking copy, no device-wide synchronisation, nothing is ever evicted
      std::vector<char> buf(map_bytes + (size_t)n_ops, 0);
      memcpy(buf.data(), level_index, (size_t)n_lev * 4);
      if (masked_levels) memcpy(buf.data() + map_bytes, masked_levels, (size_t)n_ops);
      SMM_HIP(hipMalloc(&d_cfg, buf.size()));
      hipError_t e = hipMemcpy(d_cfg, buf.data(), buf.size(), hipMemcpyHostToDevice);
      if (e != hipSuccess) {
        (void)hipFree(d_cfg);
        (void)hipGetLastError();
        return fail(SMM_ERR_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
      }
      g->cfg_cache.emplace(std::move(key), d_cfg);
    }
  }
  *d_map = (const int32_t*)d_cfg;
  *d_masked = masked_levels ? (const uint8_t*)d_cfg + map_bytes : nullptr;
  return SMM_OK;
}
}  // extern "C++"

static int smm_group_prepare_impl(smm_group_t g, int64_t n_lev, const int32_t* level_index,
                      const uint8_t* masked_levels) {
  if (!g) return fail(SMM_ERR_INVALID, "null group");
  DeviceGuard guard(g->device);
  if (!guard.ok) return fail(SMM_ERR_HIP, "cannot select the group's device");
  const int32_t* d_map;
  const uint8_t* d_masked;
  return group_level_cfg(g, n_lev, level_index, masked_levels, 0.0, 0u, &d_map, &d_masked);
}

static int smm_group_apply_impl(smm_group_t g, const void* x, int x_dtype, int64_t xs_outer, int64_t xs_lev,
                    int64_t xs_inner, void* y, int y_dtype, int64_t ys_outer, int64_t ys_lev,
                    int64_t ys_inner, int64_t n_outer, int64_t n_lev, int64_t n_inner,
                    const int32_t* level_index, const uint8_t* masked_levels, double remap_area_min,
                    unsigned flags, void* stream) {
  if (int frc = check_flags(flags)) return frc;
  if (!g) return fail(SMM_ERR_INVALID, "null group");
  DeviceGuard guard(g->device);
  if (!guard.ok) return fail(SMM_ERR_HIP, "cannot select the group's device");
  const int32_t* d_map;
  const uint8_t* d_masked;
  int rc = group_level_cfg(g, n_lev, level_index, masked_levels, remap_area_min, flags, &d_map, &d_masked);
  if (rc || n_lev == 0) return rc;
  const smm_operator* op0 = g->ops[0];
  return run_apply(g->d_descs, d_map, d_masked, op0->csr.n_src, op0->csr.n_dst, g->tile_which,
                   g->tile_valid, g->tile_preferred, (g->tile_reuse ? 1 : 0), g->tile_max_chunks, g->max_row_nnz, x, x_dtype, xs_outer, xs_lev, xs_inner, y,
                   y_dtype, ys_outer, ys_lev, ys_inner, n_outer, n_lev, n_inner, remap_area_min,
                   flags, (hipStream_t)stream);
}

extern "C++" {
// What smm_apply_sb would reject for one of the selected levels, checked for all of them up front.
static int check_sb_levels(smm_group_t g, int64_t n_lev, const int32_t* level_index, const uint8_t* masked_levels,
                           double remap_area_min, unsigned flags) {
  if (!(remap_area_min >= 0.0 && remap_area_min <= 1.0))
    return fail(SMM_ERR_INVALID, "remap_area_min must be within [0, 1]");  // regrid.py:124-125
  const int n_ops = (int)g->ops.size();
  for (int64_t l = 0; l < n_lev; ++l) {
    const int32_t w = level_index[l];
    if (w < 0 || w >= n_ops)
      return fail(SMM_ERR_INVALID, "level_index[" + std::to_string(l) + "]=" + std::to_string(w) + " outside the group");
    const smm_operator* op = g->ops[(size_t)w];
    const bool m = (flags & SMM_APPLY_MASKED) && (!masked_levels || masked_levels[w]);
    if (m && !op->d_imask) return fail(SMM_ERR_INVALID, "masked apply requested but a level has no dst_imask");
    if (remap_area_min > 0.0 && !op->d_frac)
      return fail(SMM_ERR_INVALID, "remap_area_min > 0 requested but a level has no dst_frac");
  }
  return SMM_OK;
}
}  // extern "C++"

static int smm_group_prepare_sb_impl(smm_group_t g) {
  if (!g) return fail(SMM_ERR_INVALID, "null group");
  DeviceGuard guard(g->device);
  if (!guard.ok) return fail(SMM_ERR_HIP, "cannot select the group's device");
  for (smm_operator* op : g->ops) {
    int rc = ensure_sb(op);
    if (rc) return rc;
  }
  return SMM_OK;
}

// All data levels in ONE launch of the batch-fastest kernel (smm_group_apply_sb_kernel): the levels' CSR / epilogue
// pointers travel by value in the kernel arguments (a pointer table in device memory made the column / weight /
// row-pointer streams vector loads, 162 instead of 88 VGPRs; through kernarg + constant-address-space views they
// stay scalar), so the dispatcher balances thin deep levels against the surface ones and there is no ramp-up and
// tail per level.  Groups of more than kSbGroupLevels data levels take several launches on the caller's stream.
// BASELINE config 3 kept batch-fastest, same box: 9.51 ms against 10.24 ms for one launch per level dealt over a
// pool of 8 streams (round 4's form, removed: profiles/r05_cfg3sb_grouped_vs_stream_pool.txt) and 14.4 ms for one
// launch per level on one stream (still there: SMM_TUNE_SB_LEVEL_LAUNCHES, and for levels beyond the grid limit).
static int smm_group_apply_sb_impl(smm_group_t g, const void* x, int x_dtype, int64_t xs_lev, int64_t ldx, void* y,
                       int y_dtype, int64_t ys_lev, int64_t ys_batch, int64_t n_batch, int64_t n_lev,
                       const int32_t* level_index, const uint8_t* masked_levels, double remap_area_min,
                       unsigned flags, void* stream) {
  if (int frc = check_flags(flags)) return frc;
  if (!g) return fail(SMM_ERR_INVALID, "null group");
  if (n_batch < 0 || n_lev < 0) return fail(SMM_ERR_INVALID, "negative batch size / level count");
  if (n_lev > 0 && !level_index) return fail(SMM_ERR_INVALID, "null level_index");
  if (flags & SMM_APPLY_SB_PACKED)
    return fail(SMM_ERR_UNSUPPORTED, "packed fields are per operator: a group takes whole (S, B) slabs");
  if ((x_dtype != SMM_F32 && x_dtype != SMM_F64) || (y_dtype != SMM_F32 && y_dtype != SMM_F64))
    return fail(SMM_ERR_UNSUPPORTED, "field dtype must be SMM_F32 or SMM_F64");
  const int n_ops = (int)g->ops.size();
  for (int64_t l = 0; l < n_lev; ++l)
    if (level_index[l] < 0 || level_index[l] >= n_ops)
      return fail(SMM_ERR_INVALID, "level_index[" + std::to_string(l) + "]=" + std::to_string(level_index[l]) +
                                       " outside the group");
  if (n_lev == 0 || n_batch == 0 || g->ops[0]->csr.n_dst == 0) return SMM_OK;
  if (!x || !y) return fail(SMM_ERR_INVALID, "null field pointer");
  const size_t xsz = x_dtype == SMM_F64 ? 8 : 4, ysz = y_dtype == SMM_F64 ? 8 : 4;
  // the whole call is validated before the first launch (as smm_group_apply does): a later level's
  // missing dst_imask / dst_frac or a bad stride must not surface after earlier levels wrote part of Y
  if (ldx < n_batch) return fail(SMM_ERR_INVALID, "ldx smaller than the batch");
  if (ys_batch < ((flags & SMM_APPLY_SB_Y_SB) ? n_batch : g->ops[0]->csr.n_dst))
    return fail(SMM_ERR_INVALID, "ys_batch smaller than a row of Y");
  if ((uintptr_t)x % xsz || (uintptr_t)y % ysz) return fail(SMM_ERR_INVALID, "field pointer is not element aligned");
  {
    int vrc = check_sb_levels(g, n_lev, level_index, masked_levels, remap_area_min, flags);
    if (vrc) return vrc;
  }
  const bool per_level_launches = smm::tuning(SMM_TUNE_SB_LEVEL_LAUNCHES) == 1;
  hipStream_t caller = (hipStream_t)stream;
  DeviceGuard guard(g->device);
  if (!guard.ok) return fail(SMM_ERR_HIP, "cannot select the group's device");
  const int64_t n_dst = g->ops[0]->csr.n_dst;
  const int64_t per_level = ((n_dst + (ysz == 8 ? 16 : 32) - 1) / (ysz == 8 ? 16 : 32)) * ((n_batch + 127) / 128);
  if (!per_level_launches && per_level <= grid_limit()) {
    // grouped launches of as many levels as the kernel arguments (and the launch grid) hold
    for (smm_operator* op : g->ops) {          // the canonical CSR copies the kernel reads (uploaded once)
      int erc = ensure_sb(op);
      if (erc) return erc;
    }
    const int64_t max_lev = std::max<int64_t>(1, std::min<int64_t>(kSbGroupLevels, grid_limit() / per_level));
    const bool fill = !(flags & SMM_APPLY_NO_FILL);
    for (int64_t l0 = 0; l0 < n_lev; l0 += max_lev) {
      SbGroupArgs a{};
      a.n_lev = (int)std::min<int64_t>(max_lev, n_lev - l0);
      a.x = (const char*)x + (size_t)l0 * xs_lev * xsz;
      a.y = (char*)y + (size_t)l0 * ys_lev * ysz;
      a.xs_lev = xs_lev;
      a.ys_lev = ys_lev;
      a.ldx = ldx;
      a.ldy = ys_batch;
      a.n_batch = n_batch;
      a.n_dst = n_dst;
      a.area_min = remap_area_min;
      for (int i = 0; i < a.n_lev; ++i) {
        const int w = level_index[l0 + i];
        const smm_operator* op = g->ops[(size_t)w];
        const bool m = (flags & SMM_APPLY_MASKED) && (!masked_levels || masked_levels[w]);   // regrid.py:405
        a.lev[i] = SbLevelPtrs{op->d_csr_rowptr, op->d_csr_col, op->d_csr_val, m ? op->d_imask : nullptr, op->d_frac};
      }
      const int rc = x_dtype == SMM_F64
                         ? (y_dtype == SMM_F64 ? launch_sb_group<double, double>(a, fill, flags, caller)
                                               : launch_sb_group<double, float>(a, fill, flags, caller))
                         : (y_dtype == SMM_F64 ? launch_sb_group<float, double>(a, fill, flags, caller)
                                               : launch_sb_group<float, float>(a, fill, flags, caller));
      if (rc) return rc;
    }
    return SMM_OK;
  }
  // A level whose own grid exceeds the launch limit (smm_apply_sb cuts its batch into parts), or the tuning knob
  // SMM_TUNE_SB_LEVEL_LAUNCHES: one launch per data level on the caller's stream.
  int status = SMM_OK;
  for (int64_t l = 0; l < n_lev && status == SMM_OK; ++l) {
    const int w = level_index[l];
    unsigned fl = flags & ~(unsigned)SMM_APPLY_MASKED;
    if ((flags & SMM_APPLY_MASKED) && (!masked_levels || masked_levels[w])) fl |= SMM_APPLY_MASKED;   // regrid.py:405
    status = smm_apply_sb(g->ops[(size_t)w], (const char*)x + (size_t)l * xs_lev * xsz, x_dtype, ldx,
                          (char*)y + (size_t)l * ys_lev * ysz, y_dtype, ys_batch, n_batch, remap_area_min, fl, caller);
  }
  return status;
}

extern "C++" {
static void fill_launch_info(const LaunchInfo& li, int* kernel, int* j_per_block, int* rows_per_step,
                             int* rows_per_block, int64_t* n_blocks, int64_t* lds_bytes, int* big_operator) {
  if (kernel) *kernel = li.tile ? (li.dma ? 2 : 1) : 0;
  if (j_per_block) *j_per_block = li.j_per_block;
  if (rows_per_step) *rows_per_step = li.rows_per_step;
  if (rows_per_block) *rows_per_block = li.rows_per_block;
  if (n_blocks) *n_blocks = li.n_blocks;
  if (lds_bytes) *lds_bytes = li.lds_bytes;
  if (big_operator) *big_operator = li.big_operator ? 1 : 0;
}
}  // extern "C++"

static int smm_operator_launch_info_impl(smm_operator_t op, int x_dtype, int64_t n_batch, unsigned flags, int* kernel,
                             int* j_per_block, int* rows_per_step, int* rows_per_block, int64_t* n_blocks,
                             int64_t* lds_bytes, int* big_operator) {
  if (int frc = check_flags(flags)) return frc;
  if (!op) return fail(SMM_ERR_INVALID, "null operator");
  const int pw = op->native_plan();
  const smm_operator::TilePlan& pl = op->plan[pw];
  LaunchInfo li;
  int rc = run_apply(op->d_desc, nullptr, nullptr, op->csr.n_src, op->csr.n_dst, pw, pl.valid, pl.preferred,
                     (pl.reuse ? 1 : 0), pl.max_chunks, op->csr.max_row_nnz, nullptr, x_dtype, op->csr.n_src, 0, 0,
                     nullptr, SMM_F64, op->csr.n_dst, 0, 0, n_batch, 1, 1, 0.0, flags, nullptr, &li);
  if (rc) return rc;
  fill_launch_info(li, kernel, j_per_block, rows_per_step, rows_per_block, n_blocks, lds_bytes, big_operator);
  return SMM_OK;
}

static int smm_group_launch_info_impl(smm_group_t g, int x_dtype, int64_t n_outer, int64_t n_lev, int64_t n_inner,
                          unsigned flags, int* kernel, int* j_per_block, int* rows_per_step,
                          int* rows_per_block, int64_t* n_blocks, int64_t* lds_bytes, int* big_operator) {
  if (int frc = check_flags(flags)) return frc;
  if (!g) return fail(SMM_ERR_INVALID, "null group");
  const smm_operator* op0 = g->ops[0];
  LaunchInfo li;
  int rc = run_apply(g->d_descs, nullptr, nullptr, op0->csr.n_src, op0->csr.n_dst, g->tile_which, g->tile_valid,
                     g->tile_preferred, (g->tile_reuse ? 1 : 0), g->tile_max_chunks, g->max_row_nnz, nullptr,
                     x_dtype, 0, 0, 0, nullptr, SMM_F64, 0, 0, 0, n_outer, n_lev, n_inner, 0.0, flags, nullptr, &li);
  if (rc) return rc;
  fill_launch_info(li, kernel, j_per_block, rows_per_step, rows_per_block, n_blocks, lds_bytes, big_operator);
  return SMM_OK;
}

// Host-buffer variant of smm_group_apply: X host (n_outer, n_lev, n_inner, S) C-contiguous,
// Y host (n_outer, n_inner, n_lev, D) when transpose != 0 (regrid.py:420-427), else
// (n_lev, n_outer, n_inner, D) (concat order, regrid.py:410).  Chunks of the outer axis
// stream through the group's double-buffered H2D / kernel / D2H pipeline.
static int smm_group_apply_host_impl(smm_group_t g, const void* x_host, int x_dtype, void* y_host, int y_dtype,
                         int64_t n_outer, int64_t n_lev, int64_t n_inner, int transpose,
                         const int32_t* level_index, const uint8_t* masked_levels,
                         double remap_area_min, unsigned flags, int64_t chunk_outer) {
  if (int frc = check_flags(flags)) return frc;
  if (!g) return fail(SMM_ERR_INVALID, "null group");
  if (n_outer < 0 || n_lev < 0 || n_inner < 0) return fail(SMM_ERR_INVALID, "negative batch size");
  if ((x_dtype != SMM_F32 && x_dtype != SMM_F64) || (y_dtype != SMM_F32 && y_dtype != SMM_F64))
    return fail(SMM_ERR_UNSUPPORTED, "field dtype must be SMM_F32 or SMM_F64");
  const int64_t S = g->ops[0]->csr.n_src, D = g->ops[0]->csr.n_dst;
  if (n_outer == 0 || n_lev == 0 || n_inner == 0 || D == 0) return SMM_OK;
  if (!x_host || !y_host) return fail(SMM_ERR_INVALID, "null field pointer");
  DeviceGuard guard(g->device);
  if (!guard.ok) return fail(SMM_ERR_HIP, "cannot select the group's device");

  const size_t xsz = x_dtype == SMM_F64 ? 8 : 4, ysz = y_dtype == SMM_F64 ? 8 : 4;
  const size_t xrow_d = (((size_t)S * xsz + 127) / 128) * 128;   // device rows start on 128-B lines (see smm_apply_host)
  const int64_t ldx_d = (int64_t)(xrow_d / xsz);
  const int64_t rows_per_outer = n_lev * n_inner;
  const size_t x_outer_d = (size_t)rows_per_outer * xrow_d;   // device bytes per outer index
  const size_t y_outer = (size_t)rows_per_outer * D * ysz;    // Y bytes per outer index
  // Packing variant (see smm_apply_host): when the selected levels use at most four fifths of their source
  // cells in total -- masked ocean levels thin out with depth -- each level's used cells of a chunk are
  // packed batch-fastest, one (U_l, batch) block per data level, and every level runs through the
  // batch-fastest kernel on its block.
  if (n_lev > 0 && !level_index) return fail(SMM_ERR_INVALID, "null level_index");
  int64_t used_total = 0;
  for (int64_t l = 0; l < n_lev; ++l) {
    if (level_index[l] < 0 || level_index[l] >= (int32_t)g->ops.size())
      return fail(SMM_ERR_INVALID, "level_index[" + std::to_string(l) + "] outside the group");
    used_total += g->ops[(size_t)level_index[l]]->csr.n_used_src;
  }
  // Chunks of the pipeline.  Whole rows: blocks of the outer axis with every level.  Packed: a chunk needs >= 32 batch
  // entries per level to feed the batch-fastest kernel and the pack loops; when that many entries of ALL selected levels
  // fit the staging budget a chunk is again a block of the outer axis (host_chunk_units); when they do not (config 3: 39 M
  // used cells per time step, 10 GB for 32 steps) the chunks become LEVEL-MAJOR: a few consecutive data levels x as many
  // outer indices as one level's used cells allow (round 6; before, such a field went whole rows over PCIe: config 3 ships
  // 106 GB that way and 37 GB packed).  One time step per chunk with all levels (round 3) ran 4x slower than whole rows.
  const int64_t min_outer = (32 + n_inner - 1) / n_inner;
  const bool may_pack = !(flags & (SMM_APPLY_HOST_NO_PACK | SMM_APPLY_KERNEL_SELL | SMM_APPLY_KERNEL_TILE)) &&
                        used_total > 0 && used_total * 5 <= n_lev * S * 4;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    free_b = 0;
  }
  // X and Y bytes per outer index size the chunk (smm_internal.h host_chunk_units, as in smm_apply_host)
  const smm::HostChunk hc = smm::host_chunk_units(n_outer, x_outer_d, y_outer,
                                                  may_pack ? (size_t)used_total * n_inner * xsz : 0, min_outer, 1,
                                                  chunk_outer, free_b);
  struct GChunk {
    int64_t o0, no, l0, nl;
    size_t x_bytes;   // packed X bytes of the chunk (pack mode)
  };
  std::vector<GChunk> chunks;
  const int64_t budget_kb = smm::tuning(SMM_TUNE_HOST_CHUNK_KB);   // tests: > 0 forces level-major chunks of that staging budget
  bool pack = hc.pack && budget_kb <= 0;
  if (pack || !may_pack || chunk_outer > 0) {
    for (int64_t o0 = 0; o0 < n_outer; o0 += hc.units)
      chunks.push_back({o0, std::min<int64_t>(hc.units, n_outer - o0), 0, n_lev, 0});
  }
  // level-major chunks pay from 8 batch entries on (12 monthly means of 75 levels: 10.6 GB whole rows, 3.7 GB packed; the
  // batch-fastest kernel fills few of its lanes then, but PCIe, not the kernel, is what such a call waits for)
  const int64_t min_outer_lm = std::max<int64_t>(1, (8 + n_inner - 1) / n_inner);
  if (!pack && may_pack && chunk_outer <= 0 && n_outer >= min_outer_lm) {
    // level-major: the staging budget per chunk (SMM_TUNE_HOST_CHUNK_KB lowers it so that tests reach every branch)
    const size_t target = budget_kb > 0 ? (size_t)budget_kb << 10 : (size_t)256 << 20, cap = 4 * target;
    int64_t max_used = 0;
    for (int64_t l = 0; l < n_lev; ++l) max_used = std::max(max_used, g->ops[(size_t)level_index[l]]->csr.n_used_src);
    const size_t per_outer = (size_t)std::max<int64_t>(max_used, 1) * n_inner * xsz;   // the widest level, one outer index
    int64_t bo = (int64_t)(cap / per_outer);
    if (free_b > 0) bo = std::min<int64_t>(bo, (int64_t)(free_b / 8 / (per_outer + (size_t)n_inner * D * ysz)));
    bo = std::min(bo, n_outer);
    if (bo >= std::min(min_outer, n_outer)) {
      pack = true;
      chunks.clear();
      for (int64_t o0 = 0; o0 < n_outer; o0 += bo) {
        const int64_t no = std::min(bo, n_outer - o0);
        for (int64_t l0 = 0; l0 < n_lev;) {
          int64_t nl = 0;
          size_t bytes = 0;
          do {
            bytes += (size_t)g->ops[(size_t)level_index[l0 + nl]]->csr.n_used_src * no * n_inner * xsz;
            ++nl;
          } while (l0 + nl < n_lev &&
                   bytes + (size_t)g->ops[(size_t)level_index[l0 + nl]]->csr.n_used_src * no * n_inner * xsz <= target);
          chunks.push_back({o0, no, l0, nl, bytes});
          l0 += nl;
        }
      }
    }
  }
  if (chunks.empty())   // packing not possible after all: whole rows
    for (int64_t o0 = 0; o0 < n_outer; o0 += hc.units)
      chunks.push_back({o0, std::min<int64_t>(hc.units, n_outer - o0), 0, n_lev, 0});
  size_t max_x = 0, max_y = 0, max_rows = 0;
  for (GChunk& c : chunks) {
    if (pack && c.x_bytes == 0)
      for (int64_t l = c.l0; l < c.l0 + c.nl; ++l)
        c.x_bytes += (size_t)g->ops[(size_t)level_index[l]]->csr.n_used_src * c.no * n_inner * xsz;
    max_x = std::max(max_x, pack ? c.x_bytes : (size_t)c.no * x_outer_d);
    max_y = std::max(max_y, (size_t)c.no * n_inner * c.nl * D * ysz);
    max_rows = std::max(max_rows, (size_t)c.no * rows_per_outer);
  }
  if (pack) {
    // every selected level is checked before the first launch (as smm_group_apply does): a level that
    // lacks dst_imask / dst_frac must not surface after earlier levels have written part of Y
    int vrc = check_sb_levels(g, n_lev, level_index, masked_levels, remap_area_min, flags);
    if (vrc) return vrc;
    int prc = smm_group_prepare_sb(g);
    if (prc) return prc;
  }
  const bool x_direct = is_pinned(x_host) && !pack, y_direct = is_pinned(y_host);

  std::lock_guard<std::mutex> pipe_lock(g->pipe_mu);
  HostPipe& pipe = g->pipe;
  SMM_HIP(pipe.ensure(max_x, max_y, x_direct ? 0 : (pack ? max_x : max_rows * S * xsz), y_direct ? 0 : max_y));

  const int64_t n_chunks = (int64_t)chunks.size();
  CallStats st;
  // Y of a chunk on the device: (no, n_inner, nl, D) when transpose, else (nl, no, n_inner, D); on the host the chunk is
  // the level range [l0, l0 + nl) of rows [o0, o0 + no): `rows` runs of nl * D values at a pitch of n_lev * D (transpose),
  // nl runs of no * n_inner * D values (else).  whole = the chunk holds every level: one contiguous block when transpose.
  auto y_to_host = [&](const GChunk& c, const char* src, bool async, hipStream_t stream) -> int {
    const int64_t bc = c.no * n_inner;
    if (transpose) {
      char* dst = (char*)y_host + ((size_t)c.o0 * n_inner * n_lev + (size_t)c.l0) * D * ysz;
      const size_t width = (size_t)c.nl * D * ysz, pitch = (size_t)n_lev * D * ysz;
      if (async) {
        if (c.nl == n_lev)
          SMM_HIP(hipMemcpyAsync(dst, src, (size_t)bc * width, hipMemcpyDeviceToHost, stream));
        else
          SMM_HIP(hipMemcpy2DAsync(dst, pitch, src, width, width, (size_t)bc, hipMemcpyDeviceToHost, stream));
      } else if (c.nl == n_lev) {
        int rc = host_copy(dst, src, (size_t)bc * width);
        if (rc) return rc;
      } else {
        int rc = stage_status(smm::host_copy_2d(dst, pitch, src, width, width, bc), "host copy");
        if (rc) return rc;
      }
    } else {  // device chunk is (nl, no, n_inner, D); host is (n_lev, n_outer, n_inner, D)
      const size_t blk = (size_t)bc * D * ysz;
      for (int64_t ll = 0; ll < c.nl; ++ll) {
        char* dst = (char*)y_host + (((size_t)(c.l0 + ll)) * n_outer + (size_t)c.o0) * n_inner * D * ysz;
        if (async) {
          SMM_HIP(hipMemcpyAsync(dst, src + (size_t)ll * blk, blk, hipMemcpyDeviceToHost, stream));
        } else {
          int rc = host_copy(dst, src + (size_t)ll * blk, blk);
          if (rc) return rc;
        }
      }
    }
    return SMM_OK;
  };
  auto drain = [&](int64_t c) -> int {
    const int b = (int)(c & 1);
    {
      StageTimer t(st.v[SMM_HOST_STAT_WAIT_MS]);
      SMM_HIP(hipStreamSynchronize(pipe.stream[b]));
    }
    st.chunk_done(pipe, b);
    if (!y_direct) {
      StageTimer t(st.v[SMM_HOST_STAT_COPY_OUT_MS]);
      int rc = y_to_host(chunks[(size_t)c], (const char*)pipe.hy[b], false, nullptr);
      if (rc) return rc;
    }
    return SMM_OK;
  };

  const int64_t fail_at = test_fail_chunk();
  auto pipeline = [&]() -> int {   // see smm_apply_host: errors drain both streams before returning
  for (int64_t c = 0; c < n_chunks; ++c) {
    const int b = (int)(c & 1);
    const GChunk& ck = chunks[(size_t)c];
    const int64_t o0 = ck.o0, no = ck.no;
    if (c >= 2) {
      int rc = drain(c - 2);
      if (rc) return rc;
    }
    if (c == fail_at) return fail(SMM_ERR_HIP, "injected failure (smm_debug_fail_at_chunk)");
    const char* xsrc = (const char*)x_host + (size_t)o0 * rows_per_outer * S * xsz;
    int rc = SMM_OK;
    if (pack) {
      const int64_t bc = no * n_inner;   // batch entries per level in this chunk: b = (o - o0) * n_inner + i
      size_t off = 0;                    // bytes
      {
        StageTimer t(st.v[SMM_HOST_STAT_STAGE_IN_MS]);
        for (int64_t l = ck.l0; l < ck.l0 + ck.nl; ++l) {
          smm_operator* op = g->ops[(size_t)level_index[l]];
          int hrc = host_pack((char*)pipe.hx[b] + off, xsrc + (size_t)l * n_inner * S * xsz, xsz, n_inner,
                              rows_per_outer * S, S, op->h_used, bc);
          if (hrc) return hrc;
          off += (size_t)op->csr.n_used_src * bc * xsz;
        }
      }
      SMM_HIP(hipEventRecord(pipe.ev[b][0], pipe.stream[b]));
      SMM_HIP(hipMemcpyAsync(pipe.dx[b], pipe.hx[b], off, hipMemcpyHostToDevice, pipe.stream[b]));
      SMM_HIP(hipEventRecord(pipe.ev[b][1], pipe.stream[b]));
      off = 0;
      for (int64_t ll = 0; ll < ck.nl && !rc; ++ll) {
        const int w = level_index[ck.l0 + ll];
        smm_operator* op = g->ops[(size_t)w];
        unsigned fl = (flags & SMM_APPLY_NO_FILL) | SMM_APPLY_SB_PACKED;
        if ((flags & SMM_APPLY_MASKED) && (!masked_levels || masked_levels[w])) fl |= SMM_APPLY_MASKED;
        // Y of the chunk: entry (b, ll, d) at (b * nl + ll) * D + d when transpose, at (ll * bc + b) * D + d else
        rc = smm_apply_sb(op, (char*)pipe.dx[b] + off, x_dtype, bc,
                          (char*)pipe.dy[b] + (size_t)ll * (transpose ? D : bc * D) * ysz, y_dtype,
                          transpose ? ck.nl * D : D, bc, remap_area_min, fl, pipe.stream[b]);
        off += (size_t)op->csr.n_used_src * bc * xsz;
      }
    } else {
      const int64_t rows = no * rows_per_outer;
      int64_t ys_o, ys_l, ys_i;
      if (transpose) {
        ys_o = n_inner * n_lev * D, ys_l = D, ys_i = n_lev * D;
      } else {
        ys_o = n_inner * D, ys_l = no * n_inner * D, ys_i = D;
      }
      const void* h2d_src = xsrc;
      if (!x_direct) {
        StageTimer t(st.v[SMM_HOST_STAT_STAGE_IN_MS]);
        int hrc = host_copy(pipe.hx[b], xsrc, (size_t)rows * S * xsz);
        if (hrc) return hrc;
        h2d_src = pipe.hx[b];
      }
      SMM_HIP(hipEventRecord(pipe.ev[b][0], pipe.stream[b]));
      SMM_HIP(hipMemcpy2DAsync(pipe.dx[b], xrow_d, h2d_src, (size_t)S * xsz, (size_t)S * xsz, (size_t)rows,
                               hipMemcpyHostToDevice, pipe.stream[b]));
      SMM_HIP(hipEventRecord(pipe.ev[b][1], pipe.stream[b]));
      rc = smm_group_apply(g, pipe.dx[b], x_dtype, rows_per_outer * ldx_d, n_inner * ldx_d, ldx_d,
                           pipe.dy[b], y_dtype, ys_o, ys_l, ys_i, no, n_lev, n_inner, level_index,
                           masked_levels, remap_area_min, flags, pipe.stream[b]);
    }
    if (rc) return rc;
    SMM_HIP(hipEventRecord(pipe.ev[b][2], pipe.stream[b]));
    if (!y_direct) {
      SMM_HIP(hipMemcpyAsync(pipe.hy[b], pipe.dy[b], (size_t)no * n_inner * ck.nl * D * ysz, hipMemcpyDeviceToHost,
                             pipe.stream[b]));
    } else {
      int yrc = y_to_host(ck, (const char*)pipe.dy[b], true, pipe.stream[b]);
      if (yrc) return yrc;
    }
    SMM_HIP(hipEventRecord(pipe.ev[b][3], pipe.stream[b]));
  }
  for (int64_t c = std::max<int64_t>(0, n_chunks - 2); c < n_chunks; ++c) {
    int rc = drain(c);
    if (rc) return rc;
  }
  return SMM_OK;
  };
  const int prc = pipeline();
  if (prc) pipe.quiesce();
  return prc;
}

}  // extern "C"

// ---- the guarded entry points: whatever an implementation above throws (std::bad_alloc from a plan vector, a
// std::system_error) becomes a status here -- nothing crosses the extern "C" boundary (include/smmregrid_amd.h)
extern "C" {

int smm_operator_set_epilogue(smm_operator_t op, const int32_t* dst_imask, const double* dst_frac) {
  return guarded([&] { return smm_operator_set_epilogue_impl(op, dst_imask, dst_frac); });
}

int smm_operator_mask_apply(smm_operator_t op, const int32_t* src_imask, int32_t* dst_imask) {
  return guarded([&] { return smm_operator_mask_apply_impl(op, src_imask, dst_imask); });
}

int smm_operator_prepare_sb(smm_operator_t op) {
  return guarded([&] { return smm_operator_prepare_sb_impl(op); });
}

int smm_operator_used_sources(smm_operator_t op, int32_t* used) {
  return guarded([&] { return smm_operator_used_sources_impl(op, used); });
}

int smm_apply(smm_operator_t op, const void* x, int x_dtype, int64_t ldx, void* y, int y_dtype, int64_t ldy, int64_t n_batch, double remap_area_min, unsigned flags, void* stream) {
  return guarded([&] { return smm_apply_impl(op, x, x_dtype, ldx, y, y_dtype, ldy, n_batch, remap_area_min, flags, stream); });
}

int smm_apply_sb(smm_operator_t op, const void* x, int x_dtype, int64_t ldx, void* y, int y_dtype, int64_t ldy, int64_t n_batch, double remap_area_min, unsigned flags, void* stream) {
  return guarded([&] { return smm_apply_sb_impl(op, x, x_dtype, ldx, y, y_dtype, ldy, n_batch, remap_area_min, flags, stream); });
}

int smm_apply_host(smm_operator_t op, const void* x_host, int x_dtype, int64_t ldx, void* y_host, int y_dtype, int64_t ldy, int64_t n_batch, double remap_area_min, unsigned flags, int64_t chunk_rows) {
  return guarded([&] { return smm_apply_host_impl(op, x_host, x_dtype, ldx, y_host, y_dtype, ldy, n_batch, remap_area_min, flags, chunk_rows); });
}

int smm_group_create(const smm_operator_t* ops, int n_ops, smm_group_t* out) {
  return guarded([&] { return smm_group_create_impl(ops, n_ops, out); });
}

int smm_group_prepare(smm_group_t g, int64_t n_lev, const int32_t* level_index, const uint8_t* masked_levels) {
  return guarded([&] { return smm_group_prepare_impl(g, n_lev, level_index, masked_levels); });
}

int smm_group_apply(smm_group_t g, const void* x, int x_dtype, int64_t xs_outer, int64_t xs_lev, int64_t xs_inner, void* y, int y_dtype, int64_t ys_outer, int64_t ys_lev, int64_t ys_inner, int64_t n_outer, int64_t n_lev, int64_t n_inner, const int32_t* level_index, const uint8_t* masked_levels, double remap_area_min, unsigned flags, void* stream) {
  return guarded([&] { return smm_group_apply_impl(g, x, x_dtype, xs_outer, xs_lev, xs_inner, y, y_dtype, ys_outer, ys_lev, ys_inner, n_outer, n_lev, n_inner, level_index, masked_levels, remap_area_min, flags, stream); });
}

int smm_group_prepare_sb(smm_group_t g) {
  return guarded([&] { return smm_group_prepare_sb_impl(g); });
}

int smm_group_apply_sb(smm_group_t g, const void* x, int x_dtype, int64_t xs_lev, int64_t ldx, void* y, int y_dtype, int64_t ys_lev, int64_t ys_batch, int64_t n_batch, int64_t n_lev, const int32_t* level_index, const uint8_t* masked_levels, double remap_area_min, unsigned flags, void* stream) {
  return guarded([&] { return smm_group_apply_sb_impl(g, x, x_dtype, xs_lev, ldx, y, y_dtype, ys_lev, ys_batch, n_batch, n_lev, level_index, masked_levels, remap_area_min, flags, stream); });
}

int smm_group_apply_host(smm_group_t g, const void* x_host, int x_dtype, void* y_host, int y_dtype, int64_t n_outer, int64_t n_lev, int64_t n_inner, int transpose, const int32_t* level_index, const uint8_t* masked_levels, double remap_area_min, unsigned flags, int64_t chunk_outer) {
  return guarded([&] { return smm_group_apply_host_impl(g, x_host, x_dtype, y_host, y_dtype, n_outer, n_lev, n_inner, transpose, level_index, masked_levels, remap_area_min, flags, chunk_outer); });
}

int smm_operator_launch_info(smm_operator_t op, int x_dtype, int64_t n_batch, unsigned flags, int* kernel, int* j_per_block, int* rows_per_step, int* rows_per_block, int64_t* n_blocks, int64_t* lds_bytes, int* big_operator) {
  return guarded([&] { return smm_operator_launch_info_impl(op, x_dtype, n_batch, flags, kernel, j_per_block, rows_per_step, rows_per_block, n_blocks, lds_bytes, big_operator); });
}

int smm_group_launch_info(smm_group_t g, int x_dtype, int64_t n_outer, int64_t n_lev, int64_t n_inner, unsigned flags, int* kernel, int* j_per_block, int* rows_per_step, int* rows_per_block, int64_t* n_blocks, int64_t* lds_bytes, int* big_operator) {
  return guarded([&] { return smm_group_launch_info_impl(g, x_dtype, n_outer, n_lev, n_inner, flags, kernel, j_per_block, rows_per_step, rows_per_block, n_blocks, lds_bytes, big_operator); });
}

int smm_operator_plan_info(smm_operator_t op, int* kernel_kind, int64_t* lds_bytes, int64_t* staged_src_elems) {
  return guarded([&] { return smm_operator_plan_info_impl(op, kernel_kind, lds_bytes, staged_src_elems); });
}

int smm_operator_export_csr(smm_operator_t op, int64_t* rowptr, int32_t* col, double* val) {
  return guarded([&] { return smm_operator_export_csr_impl(op, rowptr, col, val); });
}

}  // extern "C"
