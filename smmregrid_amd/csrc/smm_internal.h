// Internal structures shared by the host builder (smm_build.cpp) and the
// device side (smm_device.hip).  Not part of the ABI.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace smm {

// Canonical CSR: row = destination cell, columns ascending, duplicates summed.
// This is the structure scipy's coo_matrix((w,(dst,src))).tocsr() +
// sum_duplicates() + sort_indices() yields, and what the reference's
// sparse.COO([src,dst], w) holds after its constructor sorted the coords
// (weights.py:37-39).
struct HostCsr {
  int64_t n_src = 0, n_dst = 0, nnz = 0;
  int64_t n_used_src = 0;   // distinct source cells carrying >= 1 link (U of SURVEY 8d)
  int64_t max_row_nnz = 0;
  std::vector<int64_t> rowptr;  // n_dst + 1
  std::vector<int32_t> col;     // nnz
  std::vector<double> val;      // nnz
};

// SELL-64: destination rows in slices of 64 consecutive rows (= one wavefront);
// inside a slice the links are stored slot-major (slot k of lane r at
// off + k*64 + r) and padded to the longest row of the slice, so a wave reads
// col/val with one coalesced 256-B / 512-B access per slot.
struct HostSell {
  int64_t n_slices = 0;
  int64_t n_slots = 0;              // total padded slots (multiple of 64)
  std::vector<int64_t> slice_off;   // n_slices + 1, element offsets (multiples of 64)
  std::vector<int32_t> rowlen;      // n_slices * 64 (rows past n_dst have length 0)
  std::vector<int32_t> col;         // n_slots (padding: 0)
  std::vector<double> val;          // n_slots (padding: 0.0)
};

// Threads of the host-side builders below.  set_host_threads(n): n > 0 fixes the count, 0 = automatic
// (hardware threads, at most 16, shared between the builders running at that moment, and never more
// than the work is worth); returns the previous setting.  The results do not depend on the count.
int set_host_threads(int n);
int fixed_host_threads();   // the count set by set_host_threads (0 = automatic)
int host_threads(int64_t work_items, int64_t min_items_per_thread);
// CPUs this process may really use: the scheduler affinity capped by the cgroup CPU quota (a 16-CPU share of a
// 256-thread host has affinity 256 and a quota of 16); >= 1, read once.
int usable_cpus();
// Test hook of the builders' worker pools (tests/cpp/build_harness.cpp): no_threads = behave as if no thread could
// be started (every task then runs on the caller); throw_in_task >= 0 = the task body started after that many
// others throws std::bad_alloc (-1 = off).  Whatever a worker throws is rethrown on the calling thread after
// every started thread has been joined.
void debug_builder_faults(bool no_threads, int64_t throw_in_task);

// ---- host side of the host-buffer pipelines (smm_hostpool.cpp).  None of these throws; the int results are
// 0 = done, 1 = out of memory inside a task, 2 = any other failure inside a task.
// One persistent worker pool per process: fn(ctx, i) for i in [0, n_tasks), claimed one by one by the calling
// thread and up to max_threads - 1 workers (started on first need, kept).  A worker that cannot be started
// costs parallelism only; jobs of concurrent callers take turns.
int pool_run(int64_t n_tasks, int max_threads, void (*fn)(void*, int64_t), void* ctx) noexcept;
int pool_workers();        // workers alive in this process (tests)
int staging_threads();     // threads a staging stage uses: set_host_threads(n) if n > 0, else min(16, usable_cpus())
// Test hooks of the pool (tests/cpp/build_harness.cpp): no_threads = behave as if no worker could be started;
// throw_in_task >= 0 = the task body started after that many others throws std::bad_alloc (-1 = off).
void debug_pool_faults(bool no_threads, int64_t throw_in_task);
// parallel memcpy (pageable <-> pinned staging): one thread tops out far below PCIe
int host_copy(void* dst, const void* src, size_t bytes) noexcept;
// the same for `height` rows of `width` bytes at row pitches dpitch / spitch (the level range of a chunk inside Y)
int host_copy_2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, int64_t height) noexcept;
// Pack of the host pipelines: out[u * rows + r] = X[entry r][used[u]] -- the used source cells of a chunk of batch
// entries, batch-fastest, ready for smm_apply_sb (SMM_APPLY_SB_PACKED).  Batch entry r is the source row at
// x + (r / n_inner) * stride_o + (r % n_inner) * stride_i (elements of xsz bytes): plain row blocks have
// n_inner >= rows and stride_i = ldx; one level of an (outer, level, inner, S) field has stride_o = n_lev * n_inner * S,
// stride_i = S.  streaming = write the packed block with non-temporal stores (a staging buffer only the DMA reads).
int host_pack(void* out, const void* x, size_t xsz, int64_t n_inner, int64_t stride_o, int64_t stride_i,
              const int32_t* used, int64_t U, int64_t rows, bool streaming) noexcept;

// Returns false and fills err on invalid input.
bool build_csr(int64_t n_src, int64_t n_dst, int64_t nnz, const int32_t* src1,
               const int32_t* dst1, const double* w, HostCsr& out, std::string& err);
// Adopts an already canonical CSR (what smm_operator_export_csr wrote): validates it (rowptr
// monotone from 0, columns strictly ascending inside a row and < n_src) instead of sorting.
bool adopt_csr(int64_t n_src, int64_t n_dst, const int64_t* rowptr, const int32_t* col,
               const double* val, HostCsr& out, std::string& err);
// Drops links whose (duplicate-summed) weight is exactly +-0.0; returns how many went.  With the 1e20
// fill every gathered value is finite, so such a link contributes +-0.0 to a sum that starts at +0.0:
// the results are bit-identical with or without it (not so under SMM_APPLY_NO_FILL with non-finite X,
// whose contract excludes them).  Bilinear weights between aligned grids are full of them: r1440x721 ->
// r360x180 keeps 1 link of 4.
int64_t prune_zero_links(HostCsr& csr);
void build_sell(const HostCsr& csr, HostSell& out);

// Source-tile plan for the LDS-staged kernel.  Destination rows are grouped in
// blocks of `rows_per_block` consecutive rows (4 SELL slices, 1 slice, or 1/2, 1/4, 1/8 of a slice
// for rows whose footprint is wide); the distinct source cells a block
// references are covered by aligned chunks of `chunk_elems` source elements
// (4 elements: 32 B of f64, one 16-B staging piece of f32 -- the L2 fetches whole 128-B lines whatever
// is staged, but fine chunks keep rows that do not start on a line boundary, and sparse stencils, from
// staging and straddling lines they do not need).  The kernel copies a block's chunks to
// LDS in list order with 16-B-per-lane coalesced loads, so LDS element
// (c*chunk_elems + e) holds source element chunk_src[c]*chunk_elems + e, and
// the block's links address LDS through lcol.
struct HostTilePlan {
  bool valid = false;
  int32_t rows_per_block = 0;
  int32_t chunk_elems = 0;
  int64_t n_blocks = 0;
  int64_t max_block_chunks = 0;       // largest chunk count of any block
  int64_t total_chunks = 0;           // sum over blocks = staged chunks per batch row
  int64_t total_lines = 0;            // sum over staged blocks of the distinct 16-element groups (128-B lines of
                                      // f64) their chunks touch: what HBM moves per batch row
  std::vector<int32_t> blk_lines;     // n_blocks: that count per block
  int64_t total_distinct = 0;         // sum over blocks of distinct source cells referenced
  int64_t distinct_chunks = 0;        // distinct source chunks over the whole operator
  int64_t direct_links = 0;           // links of blocks too wide to stage (gathered from X directly)
  std::vector<uint8_t> blk_direct;    // n_blocks: 1 = footprint beyond the LDS budget, no chunks listed
  std::vector<int64_t> blk_chunk_off; // n_blocks + 1 -> index into chunk_src
  std::vector<int32_t> chunk_src;     // source chunk index (element = idx * chunk_elems)
  std::vector<int32_t> lcol;          // per SELL slot: LDS element index (layout of HostSell.col)
};
// Blocks needing more than max_chunks_per_block chunks are marked direct; plan.valid == false when
// they carry more than a quarter of the links (then the SELL kernel serves the operator better).
// A few very wide blocks (polar caps of HEALPix targets) would size the LDS tile -- and with it the
// workgroups per CU -- for every block.  Picks the smallest of full_budget/8, /4, /2 chunks that
// leaves at most 1 % of the links in over-budget blocks, marks those blocks direct and drops their
// chunk lists.  Returns the budget in force (full_budget when nothing changed).
int64_t tighten_tile_plan(const HostCsr& csr, HostTilePlan& plan, int64_t full_budget);
void build_tile_plan(const HostCsr& csr, const HostSell& sell, int32_t rows_per_block,
                     int32_t chunk_elems, int64_t max_chunks_per_block, HostTilePlan& plan);

// Launch grids are 1-D and hold at most 2^31 - 1 workgroups.  A batch of (n_outer x n_inner) rows whose
// grid would be larger is cut into parts -- halves of the outer range first, then of the inner range --
// until every part fits; `blocks(n_o, n_i)` is the grid a part of that shape needs (it depends on the
// part: short batches walk fewer rows per workgroup), `emit(o0, n_o, i0, n_i)` launches one part and
// returns 0 on success.  Returns 0, emit's first non-zero status, or -1 when a single batch row alone
// exceeds the limit (nothing is launched for that part; earlier parts have been).
template <typename Blocks, typename Emit>
int split_batch(int64_t o0, int64_t n_o, int64_t i0, int64_t n_i, int64_t limit, Blocks&& blocks, Emit&& emit) {
  if (n_o <= 0 || n_i <= 0) return 0;
  if (blocks(n_o, n_i) <= limit) return emit(o0, n_o, i0, n_i);
  if (n_o > 1) {
    const int64_t h = (n_o + 1) / 2;
    if (int rc = split_batch(o0, h, i0, n_i, limit, blocks, emit)) return rc;
    return split_batch(o0 + h, n_o - h, i0, n_i, limit, blocks, emit);
  }
  if (n_i > 1) {
    const int64_t h = (n_i + 1) / 2;
    if (int rc = split_batch(o0, n_o, i0, h, limit, blocks, emit)) return rc;
    return split_batch(o0, n_o, i0 + h, n_i - h, limit, blocks, emit);
  }
  return -1;
}

// Chunk sizing of the host-buffer pipelines (smm_apply_host / smm_group_apply_host).  A chunk's X
// AND Y staging are each allocated twice on the device and twice as pinned host memory, so a chunk
// is sized from the bytes one batch unit (a row; an outer index of a level group) needs on both
// sides -- an operator with few used source cells and a large target (U << D) is bound by its Y.
struct HostChunk {
  int64_t units = 0;   // batch rows (or outer indices) per chunk, >= 1
  bool pack = false;   // the packing variant is in force for this call
};
// x_unit / y_unit: device bytes per unit of the whole-row form; xp_unit: packed X bytes per unit
// (0 = packing not applicable).  min_pack_units: fewest units per chunk that still feed the pack loops
// and the batch-fastest kernel (32 batch entries); pack_align: packed chunks are whole multiples of
// this when they are at least that long (the kernel's 128-entry batch tile).  requested > 0 = the
// caller's chunk size (kept; packing is dropped if it is below min_pack_units and does not cover
// the batch).  free_bytes = free device memory (0 = unknown): a chunk never asks for more than a
// quarter of it across its four device buffers.
HostChunk host_chunk_units(int64_t n_units, size_t x_unit, size_t y_unit, size_t xp_unit,
                           int64_t min_pack_units, int64_t pack_align, int64_t requested,
                           size_t free_bytes);

}  // namespace smm
