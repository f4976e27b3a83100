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

// Destination slot order.  The device structures (SELL slices, tile-plan blocks) are built over
// "slots": by default slot == destination row; for a 2-D destination grid whose rows are heavy the
// slots follow PATCHES of `patch_rows` grid rows x 64 grid columns, one grid row of a patch per SELL
// slice (= wavefront), so that the `patch_rows` waves of a workgroup share ONE staged source tile
// whose halo -- the source rows between vertically adjacent destination rows -- is fetched once
// instead of once per wave.  Patches at the grid's right / upper edge are padded with empty slots
// (row_of_slot == -1).
struct SlotMap {
  bool identity = true;
  int64_t n_slots = 0;
  std::vector<int32_t> row_of_slot;   // [n_slots] destination row, or -1
};
void build_patch_slots(int64_t nx, int64_t ny, int patch_rows, SlotMap& out);
// CSR whose row s is row row_of_slot[s] of `csr` (empty for padding slots); n_dst = n_slots.
void permute_csr(const HostCsr& csr, const SlotMap& slots, HostCsr& out);

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
// (one chunk = one 128-B line of f64).  The kernel copies a block's chunks to
// LDS in list order with 16-B-per-lane coalesced loads, so LDS element
// (c*chunk_elems + e) holds source element chunk_src[c]*chunk_elems + e, and
// the block's links address LDS through lcol.
struct HostTilePlan {
  bool valid = false;
  int32_t rows_per_block = 0;
  int32_t chunk_elems = 0;
  int64_t n_blocks = 0;
  int64_t max_block_chunks = 0;       // largest chunk count of any block
  int64_t total_chunks = 0;           // sum over blocks = staged lines per batch row
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

}  // namespace smm
