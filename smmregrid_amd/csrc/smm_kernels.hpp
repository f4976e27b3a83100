// Device-side structures and HIP kernels of libsmmregrid_hip (gfx950).
// Included once by smm_device.hip, which owns the handles, launch logic and the C ABI.
//
// Hot path of jhardenberg/smmregrid rebuilt for MI355X:
//   regrid.py:545-547  fill of non-finite source values with 1e20     (fused, on load)
//   regrid.py:550      tensordot(X(B,S), W(S,D))                      (CSR SpMM, HBM-bound)
//   regrid.py:553-570  dst_imask / dst_frac / >1e19 -> NaN            (fused, on store)
//   regrid.py:387-418  per-level loop, concat, transpose              (one grouped launch)
//   weights.py:47-52   mask pre-compute                               (same kernel, B = 1)
//
// Summation order: links of a destination row are accumulated sequentially in
// ascending source index with separate multiply and add (no FMA contraction),
// exactly the order of the CPU oracle (oracle/), so f64 results are bit
// identical to it.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#pragma clang fp contract(off)

// argument structs: external linkage (they appear in the launch templates' signatures, which are
// instantiated in separate translation units)

// ------------------------------------------------------------------ device structs

struct LevelDesc {
  const int64_t* slice_off;  // [n_slices + 1]
  const int32_t* col;        // SELL slots (source cell index)
  const double* val;         // SELL slots
  const int32_t* rowlen;     // [n_slices * 64]
  const uint8_t* imask;      // [n_dst] or null
  const double* frac;        // [n_dst] or null
  // LDS source-tile plan (null when not planned)
  const int64_t* blk_chunk_off;  // [n_blocks + 1]
  const int32_t* chunk_src;      // source chunk index per staged chunk
  const int32_t* lcol;           // SELL slots (LDS element index)
  const uint8_t* blk_direct;     // [n_blocks] 1 = block not staged, links gathered from X directly
};

struct ApplyArgs {
  const LevelDesc* descs;     // device array
  const int32_t* lev_map;     // device [n_lev] -> desc index, null = identity 0
  const uint8_t* lev_masked;  // device [n_descs] per-desc mask switch, null = all
  const void* x;
  void* y;
  int64_t xs_o, xs_l, xs_i;
  int64_t ys_o, ys_l, ys_i;
  int64_t n_j;       // n_outer * n_inner batch rows per level
  int64_t n_inner;
  int64_t n_dblocks; // destination blocks (4 slices each)
  int64_t n_jtiles;
  int64_t n_src, n_dst;
  double area_min;
  int masked;
  int j_per_block;   // tile kernel: batch rows walked by one workgroup
  int xcd_remap;     // tile kernel: > 0 = length of the runs of consecutive blocks given to one XCD
  int64_t n_blocks;  // grid size (for the remap)
  int tile_bytes;    // tile kernel with R > 1: LDS bytes of one batch row's tile
  int sub_shift;     // single-wave tile kernel: a block owns 64 >> sub_shift rows of its slice
};

// arguments of the batch-fastest kernel (kernel C below)
struct SbArgs {
  const int64_t* rowptr;   // [n_dst + 1] canonical CSR
  const int32_t* col;      // [nnz] source cell, or its rank among the used cells (packed X)
  const double* val;       // [nnz]
  const uint8_t* imask;    // [n_dst] or null
  const double* frac;      // [n_dst] or null
  const void* x;           // (n_rows_x, ldx): row = source cell, batch entry fastest
  void* y;                 // batch entry b of destination cell d at y + b * ldy + d  (YSB: at y + d * ldy + b)
  int64_t ldx, ldy, n_batch, n_dst;
  int64_t n_dtiles, n_btiles, n_blocks;
  double area_min;
  int masked;
  int xcd_remap;
  int b_fastest;           // > 0: width (destination tiles) of the strips the tiles are ordered in
};

// Arguments of the batch-fastest kernel over a whole level group in ONE launch (smm_group_apply_sb).  The levels'
// CSR / epilogue pointers travel BY VALUE in the kernel argument segment: a workgroup picks its level's five
// pointers with a wave-uniform index -- scalar loads from the kernarg segment -- and reads the CSR through
// constant-address-space views of them, so the column / weight / row-pointer streams stay scalar loads exactly as
// in the single-operator kernel (a pointer table in device memory turned them into vector loads: 162 instead of
// 88 VGPRs).  kSbGroupLevels * 40 B + the rest stays below the 4-KiB kernarg limit; longer groups take several launches.
constexpr int kSbGroupLevels = 88;
struct SbLevelPtrs {
  const int64_t* rowptr;
  const int32_t* col;
  const double* val;
  const uint8_t* imask;   // null = no mask applied on this level
  const double* frac;
};
struct SbGroupArgs {
  const void* x;             // level l's (S, ldx) slab at x + l * xs_lev elements
  void* y;                   // level l's results at y + l * ys_lev elements
  int64_t xs_lev, ys_lev;
  int64_t ldx, ldy, n_batch, n_dst;
  int64_t n_dtiles, n_btiles;
  int64_t blocks_per_level;  // n_dtiles * n_btiles
  double area_min;
  int xcd_remap, b_fastest;
  int n_lev;
  SbLevelPtrs lev[kSbGroupLevels];
};
static_assert(sizeof(SbGroupArgs) <= 4096, "kernel arguments are limited to 4 KiB");

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // 16-B staging piece
typedef u32x4 u32x4_t;
constexpr int kWavesPerBlock = 4;
constexpr int kThreads = kWavesPerBlock * 64;
// Granularity of the tile plan's staged chunks.  4 elements = 32 B of f64, one 16-B staging piece of f32.
// Measured against 16 (whole 128-B lines of f64), same box: config 3 with its rows packed back to back
// (S * 8 B = 80 mod 128: every row starts mid-line) 12.84 -> 11.49 ms, on 128-B lines 11.45 -> 11.26,
// config 2 2.88 -> 2.69, the others level; 2 elements (f64 only) another 2 % on packed rows.
#ifndef SMM_CHUNK_ELEMS
#define SMM_CHUNK_ELEMS 4
#endif
constexpr int kChunkElems = SMM_CHUNK_ELEMS;
static_assert(SMM_CHUNK_ELEMS >= 4 && (SMM_CHUNK_ELEMS & (SMM_CHUNK_ELEMS - 1)) == 0, "a chunk holds at least one 16-B piece of f32");
constexpr int64_t kTileMaxChunks = 8192 / SMM_CHUNK_ELEMS;  // 64 KiB of f64 per staged batch row

template <typename T>
__device__ __forceinline__ double load_fixed(const T* __restrict__ p, bool fill) {
  const T v = *p;
  // numpy.ma.fix_invalid + filled (regrid.py:545-547): the fill value is the
  // dtype's own cast of 1e20 (float32(1e20) for an f32 field).
  const T f = (T)1e20;
  return (double)((fill && !__builtin_isfinite(v)) ? f : v);
}

// The same fill applied to a whole 16-B staging piece on its way into LDS: the tile then holds
// finite values and the link loop gathers without a test per link (a 48-link row tests 48 gathered
// values per batch row, its 15 staging pieces hold 30).
template <typename XT>
__device__ __forceinline__ u32x4_t fix_piece(u32x4_t piece) {
  constexpr int N = 16 / (int)sizeof(XT);
  XT e[N];
  __builtin_memcpy(e, &piece, 16);
#pragma unroll
  for (int q = 0; q < N; ++q) e[q] = __builtin_isfinite(e[q]) ? e[q] : (XT)1e20;
  __builtin_memcpy(&piece, e, 16);
  return piece;
}

__device__ __forceinline__ double epilogue(double v, bool dead) {
  // regrid.py:559, :563-565, :570 -- every branch yields NaN, so the order is immaterial
  return (dead || v > 1e19) ? __builtin_nan("") : v;
}

// Row pointers of batch row j of level l.
__device__ __forceinline__ int64_t row_off(int64_t j, int64_t l, int64_t n_inner, int64_t s_o,
                                           int64_t s_l, int64_t s_i) {
  const int64_t o = j / n_inner, i = j - o * n_inner;
  return o * s_o + l * s_l + i * s_i;
}

// Walks batch rows j = j0, j0+1, ... of level l without a division per row.
struct RowWalker {
  int64_t off, i, n_inner, step_i, step_o;
  __device__ __forceinline__ RowWalker(int64_t j0, int64_t l, int64_t n_inner_, int64_t s_o, int64_t s_l,
                                       int64_t s_i)
      : n_inner(n_inner_), step_i(s_i), step_o(s_o - (n_inner_ - 1) * s_i) {
    const int64_t o = j0 / n_inner_;
    i = j0 - o * n_inner_;
    off = o * s_o + l * s_l + i * s_i;
  }
  __device__ __forceinline__ void next() {
    if (++i == n_inner) {
      i = 0;
      off += step_o;
    } else {
      off += step_i;
    }
  }
};

// ------------------------------------------------------------------ kernel A
// SELL-64, one destination row per lane, BT batch rows register-blocked so the
// col/val stream is read once per BT outputs and BT independent gathers are in
// flight per link.  Gathers hit X directly: neighbouring lanes read
// neighbouring source cells, L1/L2 absorb the line reuse.
template <typename XT, typename YT, int BT>
__global__ __launch_bounds__(kThreads) void smm_apply_sell_kernel(ApplyArgs a, bool fill) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  int64_t bid = blockIdx.x;
  const int64_t db = bid % a.n_dblocks;
  bid /= a.n_dblocks;
  const int64_t jt = bid % a.n_jtiles;
  const int64_t l = bid / a.n_jtiles;
  const int di = a.lev_map ? a.lev_map[l] : 0;
  const LevelDesc L = a.descs[di];

  const int64_t slice = db * kWavesPerBlock + wave;
  const int64_t d = slice * 64 + lane;
  if (slice * 64 >= a.n_dst) return;

  const int64_t j0 = jt * BT;
  const XT* __restrict__ xr[BT];
  YT* __restrict__ yr[BT];
#pragma unroll
  for (int t = 0; t < BT; ++t) {
    int64_t j = j0 + t;
    if (j > a.n_j - 1) j = a.n_j - 1;
    xr[t] = (const XT*)a.x + row_off(j, l, a.n_inner, a.xs_o, a.xs_l, a.xs_i);
    yr[t] = (YT*)a.y + row_off(j, l, a.n_inner, a.ys_o, a.ys_l, a.ys_i);
  }

  const int64_t off = L.slice_off[slice];
  const int nslots = (int)((L.slice_off[slice + 1] - off) >> 6);
  const int len = L.rowlen[d];
  const int32_t* __restrict__ cp = L.col + off + lane;
  const double* __restrict__ vp = L.val + off + lane;

  double acc[BT];
#pragma unroll
  for (int t = 0; t < BT; ++t) acc[t] = 0.0;

  // Padded slots carry a valid column (the row's last one) and weight +0.0: the running sum is
  // never -0.0 and the gathered value is finite after the fill, so acc + 0*x == acc bit for bit --
  // the loop body is branch- and select-free and two slots are in flight.
#pragma unroll 2
  for (int k = 0; k < nslots; ++k) {
    const int32_t c = cp[(int64_t)k * 64];
    const double w = vp[(int64_t)k * 64];
    double xv[BT];
#pragma unroll
    for (int t = 0; t < BT; ++t) xv[t] = load_fixed(xr[t] + c, fill);
#pragma unroll
    for (int t = 0; t < BT; ++t) {
      const double p = w * xv[t];
      acc[t] = acc[t] + p;
    }
  }
  if (len == 0) {  // a row without links never looks at X (its padded slots read column 0)
#pragma unroll
    for (int t = 0; t < BT; ++t) acc[t] = 0.0;
  }

  if (d < a.n_dst) {
    const bool use_mask = a.masked && (a.lev_masked ? a.lev_masked[di] != 0 : true);
    bool dead = false;
    if (use_mask && L.imask) dead = (L.imask[d] == 0);
    if (a.area_min > 0.0 && L.frac) dead = dead || (L.frac[d] < a.area_min);
#pragma unroll
    for (int t = 0; t < BT; ++t) {
      if (j0 + t < a.n_j) yr[t][d] = (YT)epilogue(acc[t], dead);
    }
  }
}

// ------------------------------------------------------------------ kernel B
// LDS source tile.  A workgroup owns 256 consecutive destination rows (4 SELL
// slices) and walks j_per_block batch rows.  For each batch row it copies the
// block's source chunks (whole 128-B lines, list order) into LDS with
// 16-B-per-lane coalesced loads -- every needed HBM line is fetched exactly
// once by a full-width access -- then each lane gathers its row's links from
// LDS.
//
// Kernel B, pipelined form.  Each thread owns up to NP staging pieces whose row
// offsets are computed once (they do not depend on the batch row), keeps the
// next batch row's pieces in registers while the current one is consumed from
// LDS, so the HBM latency of row j+1 overlaps the gather/store of row j and no
// index load sits in front of a data load.  Every global / LDS load in the loop
// is unconditional (clamped address + select): a load inside a per-lane branch
// makes hipcc wait for it before the next one, which serialises the memory
// round trips.  NT bit 0: non-temporal X loads (only when no staged line is
// shared between blocks), bit 1: non-temporal Y stores.
typedef u32x4 u32x4_u __attribute__((aligned(4)));  // 16-B piece that may start on any element

// Rows with more than 16 links run one wave (64 destination rows) per workgroup: their source
// tiles are large, and single-wave workgroups need no workgroup barrier, so the waves of a CU
// drift apart and overlap each other's HBM waits.
constexpr int tile_waves(int maxk) { return (maxk > 0 && maxk <= 16) ? kWavesPerBlock : 1; }

//
// R > 1 (small tiles only): R batch rows are staged, consumed and stored per barrier pair, each in
// its own LDS region.  A workgroup whose tile needs one or two pieces per thread is bound by the
// memory round trip per batch row, not by bandwidth; R rows in flight per workgroup hide it.
//
// SPLIT (single-wave blocks of 64 >> sub_shift rows, rows of up to (1 << sub_shift) * MAXK links):
// the idle lanes take over parts of the rows.  Lane (g, r) keeps links [g*Kg, (g+1)*Kg) of row r in
// registers; per batch row the G = 1 << sub_shift lane groups run one after the other, each starting
// from the sum its predecessor handed over (a shuffle), so every row is still accumulated link by
// link in ascending source order -- bit-identical -- while no link is re-read from L2.
// DMA: the staging pieces go HBM -> LDS directly (global_load_lds_dwordx4, the gfx950 LDS-DMA path) into
// a ring of two tile slots instead of HBM -> VGPR -> ds_write: row j + 1 lands in the other slot while
// row j is consumed.  No prefetch registers, one barrier per batch row, the fill test moves to the
// gather -- at twice the LDS per workgroup.  That trade pays exactly where LDS is NOT what limits the
// workgroups per CU: small tiles of the 4-wave shape (HEALPix targets: two slots of <= 8 KB keep 8
// workgroups = 32 waves per CU; config-4 geometry 5.44 -> 4.81 ms against four-row register steps).
// Larger tiles lose workgroups to the second slot (config 2: 2.71 -> 2.99 ms; the 64-row single-wave
// kernels of config 3: 12.4 -> 15.1 ms, 15-KiB tiles: 5 instead of 8 waves per CU -- the register file
// is the bigger store on this chip, 512 KB against 160 KB per CU, and the prefetched row lives there).
// Rings of 3 / 4 slots with counted s_waitcnt vmcnt(N) and a raw s_barrier were built and measured:
// never better than two slots (config-4 geometry 4.83 ms with three slots and the wait counted so that
// neither the younger row nor the latest Y store is waited for, against 4.86; 5.56 with four): that
// kernel is bound by its bytes, not by a workgroup's round trips, and deeper rings cost workgroups per CU.
template <typename XT, typename YT, int MAXK, int NP, int NT, int R = 1, bool SPLIT = false, bool DMA = false>
__global__ __launch_bounds__(tile_waves(MAXK) * 64, 2) void smm_apply_tile2_kernel(ApplyArgs a, bool fill) {
  constexpr int WPB = tile_waves(MAXK);
  constexpr int T = WPB * 64;
  static_assert(!DMA || (!SPLIT && MAXK > 0 && (R == 1 || MAXK <= 16)), "LDS-DMA staging: links in registers; multi-row steps for the 4-wave shape");
  // Where the 1e20 fill happens: rows of more than 16 links test the staging pieces on their way into
  // LDS (a 48-link row would test 48 gathered values per batch row, its ~15 pieces hold 30); short
  // rows gather few values from comparatively many staged ones (config 4: 4 links, 8 staged f32 per
  // lane and step) and test what they gather.
  constexpr bool kFixAtStage = !DMA && (MAXK == 0 || MAXK > 16);   // LDS-DMA bypasses the registers: test at the gather
  static_assert(!SPLIT || (WPB == 1 && R == 1 && MAXK > 0), "split rows: single-wave, single-row steps");
  static_assert(R == 1 || (MAXK > 0 && MAXK <= 16), "multi-row steps exist for the 4-wave shape only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  int64_t bid = blockIdx.x;
  if (a.xcd_remap > 0) {
    // Hardware block b runs on XCD b % 8.  Runs of C consecutive logical blocks (neighbours in
    // space, sharing halo lines) go to one XCD, and the runs are dealt round-robin so that every
    // XCD sees the same mix of light and heavy levels.  Blocks past the last full round keep
    // their id (bijective for any grid size).
    const int64_t C = a.xcd_remap, round = 8 * C;
    if (bid < (a.n_blocks / round) * round) {
      const int64_t xcd = bid & 7, slot = bid >> 3;
      bid = (slot / C) * round + xcd * C + (slot % C);
    }
  }
  const int64_t db = bid % a.n_dblocks;
  bid /= a.n_dblocks;
  const int64_t jt = bid % a.n_jtiles;
  const int64_t l = bid / a.n_jtiles;
  const int di = a.lev_map ? a.lev_map[l] : 0;
  const LevelDesc L = a.descs[di];

  // 4-wave blocks own 4 slices; single-wave blocks own a slice or the 64 >> sub_shift rows of it
  // whose lanes are `sub` (the other lanes idle: len 0, no store)
  const int64_t slice = WPB == 1 ? (db >> a.sub_shift) : db * WPB + wave;
  const int sub = (int)(db & ((1 << a.sub_shift) - 1));
  const int grp = lane >> (6 - a.sub_shift);            // lane group (SPLIT: part of the row's links)
  const int n_grp = 1 << a.sub_shift;
  // lane of the slice whose row this lane works on
  const int rowlane = SPLIT ? (sub << (6 - a.sub_shift)) + (lane & ((64 >> a.sub_shift) - 1)) : lane;
  const bool in_blk = SPLIT || WPB != 1 || grp == sub;
  const int64_t d = slice * 64 + rowlane;      // destination cell
  const int64_t dy = d;
  // SPLIT: the last lane group ends up with the rows' sums and stores them
  const bool row_live = in_blk && d < a.n_dst && (!SPLIT || grp == n_grp - 1);
  const bool slice_live = slice * 64 < a.n_dst;

  // MAXK > 0: the row's links live in registers across batch rows (LDS byte offsets are
  // < 65536, two per register); MAXK == 0 (rows longer than 48 links): re-read per row.
  constexpr int KREG = MAXK > 0 ? MAXK : 2;
  int len = 0, nslots = 0;
  uint32_t lc2[KREG / 2];
  double w[KREG];
  int64_t soff = 0;
  int first = 0;   // SPLIT: first slot of this lane's part of the row
  if (slice_live) {
    soff = L.slice_off[slice];
    nslots = (int)((L.slice_off[slice + 1] - soff) >> 6);
    len = (in_blk && d < a.n_dst) ? L.rowlen[d] : 0;
    if (SPLIT) {
      int longest = len;   // longest row of the block -> links per lane group (wave-uniform)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) longest = max(longest, __shfl_xor(longest, off));
      const int per_grp = (__builtin_amdgcn_readfirstlane(longest) + n_grp - 1) >> a.sub_shift;
      first = grp * per_grp;
      len = min(max(len - first, 0), per_grp);
    }
  }
  const int32_t* __restrict__ cp = L.lcol + soff + rowlane;   // global pointers on every path
  const double* __restrict__ vp = L.val + soff + rowlane;
  if (MAXK > 0 && nslots > 0) {  // wave-uniform; loads unconditional, slots past the slice clamp
    const uint32_t cpad = len > 0 ? (uint32_t)cp[(int64_t)min(first, nslots - 1) * 64] : 0u;
#pragma unroll
    for (int k = 0; k < KREG; k += 2) {
      const int k0 = min(first + k, nslots - 1), k1 = min(first + k + 1, nslots - 1);
      const uint32_t c0 = (uint32_t)cp[(int64_t)k0 * 64];
      const uint32_t c1 = (uint32_t)cp[(int64_t)k1 * 64];
      // two 16-bit LDS BYTE offsets per register (tiles are <= 64 KiB)
      lc2[k / 2] = ((k < len ? c0 : cpad) | ((k + 1 < len ? c1 : cpad) << 16)) * (uint32_t)sizeof(XT);
      // Slots past the row's length get weight +0.0 and the LDS index of the row's first link: the
      // running sum is never -0.0 (it starts at +0.0) and the staged value is finite after the fill,
      // so acc + 0*x == acc bit for bit and the walk loop needs no per-link select.  (Without the
      // fill, SMM_APPLY_NO_FILL, a NaN there belongs to the row anyway.)
      const double w0 = vp[(int64_t)k0 * 64], w1 = vp[(int64_t)k1 * 64];
      w[k] = k < len ? w0 : 0.0;
      w[k + 1] = k + 1 < len ? w1 : 0.0;
    }
  } else {
#pragma unroll
    for (int k = 0; k < KREG; k += 2) {
      lc2[k / 2] = 0u;
      w[k] = 0.0;
      w[k + 1] = 0.0;
    }
  }
  // wave-uniform trip count: longest row of this wave
  int wmax = len;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off));
  wmax = __builtin_amdgcn_readfirstlane(wmax);

  bool dead = false;
  if (row_live) {
    const bool use_mask = a.masked && (a.lev_masked ? a.lev_masked[di] != 0 : true);
    if (use_mask && L.imask) dead = (L.imask[d] == 0);
    if (a.area_min > 0.0 && L.frac) dead = dead || (L.frac[d] < a.area_min);
  }

  const int64_t c0 = L.blk_chunk_off[db];
  const int nch = (int)(L.blk_chunk_off[db + 1] - c0);
  const int32_t* __restrict__ chunk_src = L.chunk_src + c0;
  constexpr int kElemsPerPiece = 16 / (int)sizeof(XT);
  constexpr int pieces_per_chunk = kChunkElems / kElemsPerPiece;
  const int npieces = nch * pieces_per_chunk;

  // Pieces [wave*64 + k*256, +64) belong to this wave in round k.  poff = element offset of
  // the piece inside a batch row (clamped so that the 16-B load stays inside the row),
  // shift = elements by which the clamp moved it (non-zero only for the row's last piece).
  // Lanes past the block's last piece (and pieces wholly beyond the row) load offset 0 of the row
  // and write their natural LDS slot, which no link refers to.
  int np_w = __builtin_amdgcn_readfirstlane(
      npieces > wave * 64 ? (npieces - wave * 64 + T - 1) / T : 0);
  int32_t poff[NP];
  unsigned shifted = 0;
  int shift_amt = 0;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int p = tid + k * T;
    poff[k] = 0;
    if (k < np_w) {
      const int pc = min(p, npieces - 1);
      const int ch = pc / pieces_per_chunk;
      const int sub = pc - ch * pieces_per_chunk;
      int64_t e0 = (int64_t)chunk_src[ch] * kChunkElems + (int64_t)sub * kElemsPerPiece;
      if (p < npieces && e0 < a.n_src) {
        if (e0 + kElemsPerPiece > a.n_src) {
          const int64_t e1 = a.n_src >= kElemsPerPiece ? a.n_src - kElemsPerPiece : 0;
          shifted |= 1u << k;
          shift_amt = (int)(e0 - e1);
          e0 = e1;
        }
        poff[k] = (int32_t)e0;
      }
    }
  }
  const bool tiny_row = a.n_src < kElemsPerPiece;  // a 16-B load would leave the row: gather directly

  const int64_t j_begin = jt * a.j_per_block;
  int64_t j_end = j_begin + a.j_per_block;
  if (j_end > a.n_j) j_end = a.n_j;
  if (j_begin >= j_end) return;

  if (L.blk_direct[db] || tiny_row) {
    // footprint beyond the LDS budget (or a source row shorter than one 16-B piece): gather this
    // block's links straight from X (SELL arrays
    // hold the global columns), one batch row at a time; no LDS, no barrier
    if (slice_live) {
      // this path is row-per-lane; in a SPLIT kernel the lanes of group `sub` take their own rows
      const int64_t dd = SPLIT ? slice * 64 + lane : d;
      const int64_t ddy = SPLIT ? dd : dy;
      const bool dlive = SPLIT ? (grp == sub && dd < a.n_dst) : row_live;
      int dlen = len, dmax = wmax;
      bool ddead = dead;
      if (SPLIT) {
        dlen = dlive ? L.rowlen[dd] : 0;
        dmax = dlen;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) dmax = max(dmax, __shfl_xor(dmax, off));
        dmax = __builtin_amdgcn_readfirstlane(dmax);
        ddead = false;
        if (dlive) {
          const bool use_mask = a.masked && (a.lev_masked ? a.lev_masked[di] != 0 : true);
          if (use_mask && L.imask) ddead = (L.imask[dd] == 0);
          if (a.area_min > 0.0 && L.frac) ddead = ddead || (L.frac[dd] < a.area_min);
        }
      }
      const int32_t* __restrict__ gcp = L.col + soff + lane;
      const double* __restrict__ gvp = L.val + soff + lane;
      RowWalker xwd(j_begin, l, a.n_inner, a.xs_o, a.xs_l, a.xs_i);
      RowWalker ywd(j_begin, l, a.n_inner, a.ys_o, a.ys_l, a.ys_i);
      for (int64_t j = j_begin; j < j_end; ++j, xwd.next(), ywd.next()) {
        const XT* __restrict__ xrow = (const XT*)a.x + xwd.off;
        double acc = 0.0;
        // eight slots at a time: their column / weight loads, then the eight gathers, are in flight
        // together (a load per link in front of its gather would serialise the round trips)
        for (int k0 = 0; k0 < dmax; k0 += 8) {
          int32_t gc[8];
          double wv[8], xv[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int kc = min(k0 + q, nslots - 1);   // padding repeats a valid column
            gc[q] = gcp[(int64_t)kc * 64];
            wv[q] = gvp[(int64_t)kc * 64];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) xv[q] = load_fixed(xrow + gc[q], fill);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const double p = wv[q] * xv[q];
            const double sum = acc + p;
            acc = (k0 + q < dlen) ? sum : acc;
          }
        }
        if (dlive) {
          YT* __restrict__ yrow = (YT*)a.y + ywd.off;
          yrow[ddy] = (YT)epilogue(acc, ddead);
        }
      }
    }
    return;
  }
  if (npieces == 0) {
    // no destination row of this block has a link (land-only block): nothing to stage,
    // no barrier needed -- every batch row gets epilogue(0)
    if (row_live) {
      const YT out = (YT)epilogue(0.0, dead);
      RowWalker yw0(j_begin, l, a.n_inner, a.ys_o, a.ys_l, a.ys_i);
      for (int64_t j = j_begin; j < j_end; ++j) {
        YT* __restrict__ yrow = (YT*)a.y + yw0.off;
        if (NT & 2)
          __builtin_nontemporal_store(out, yrow + dy);
        else
          yrow[dy] = out;
        yw0.next();
      }
    }
    return;
  }

  if constexpr (DMA) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int tile_bytes = a.tile_bytes;   // one slot of the ring
    // piece k of this wave lands at (wave * 64 + k * T) * 16 + lane * 16 of the slot: the LDS-DMA
    // destination is a wave-uniform base plus lane * 16, the source address is per lane
    auto issue_row = [&](int64_t xoff, int slot) {
      const XT* __restrict__ xrow = (const XT*)a.x + xoff;
      char* base = smem + (size_t)slot * tile_bytes + (size_t)wave * 64 * 16;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        if (k >= np_w) break;
        __builtin_amdgcn_global_load_lds((gptr_t)(xrow + poff[k]), (lptr_t)(base + (size_t)k * T * 16), 16, 0,
                                         (NT & 1) ? 2 : 0);
      }
    };
    // A step covers R batch rows (R > 1: small tiles whose round trip, not their bytes, bounds a step): the
    // ring holds two groups of R slots, rows jb + R .. jb + 2R - 1 land in one group while rows
    // jb .. jb + R - 1 are consumed from the other.
    RowWalker xw(j_begin, l, a.n_inner, a.xs_o, a.xs_l, a.xs_i);   // next row to issue
    RowWalker yw(j_begin, l, a.n_inner, a.ys_o, a.ys_l, a.ys_i);   // next row to compute
    YT pend_out[R];
    int64_t pend_off[R];
    int n_pend = 0;
    auto flush_pending = [&]() {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r < n_pend) {
          YT* __restrict__ yrow = (YT*)a.y + pend_off[r];
          if (NT & 2)
            __builtin_nontemporal_store(pend_out[r], yrow + dy);
          else
            yrow[dy] = pend_out[r];
        }
      }
    };
    int64_t j_issue = j_begin;
    auto issue_group = [&](int group) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (j_issue < j_end) {
          issue_row(xw.off, group * R + r);
          ++j_issue;
          if (j_issue < j_end) xw.next();
        }
      }
    };
    issue_group(0);
    int group = 0;
    for (int64_t jb = j_begin; jb < j_end; jb += R) {
      asm volatile("" : "+s"(np_w), "+s"(wmax));
      if (MAXK > 16) {
#pragma unroll
        for (int q = 0; q < KREG / 2; ++q) asm volatile("" : "+v"(lc2[q]));
      }
      // this step's rows have landed (and every wave is done with the other group, about to be refilled)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (WPB > 1) __syncthreads();
      if (row_live) flush_pending();   // the previous step's results: issued before the next DMAs
      issue_group(group ^ 1);
      n_pend = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (jb + r < j_end) {
          if (slice_live) {
            const char* lds_b = smem + (size_t)(group * R + r) * tile_bytes;
            double acc = 0.0;
#ifdef SMM_EXP_SKIP_COMPUTE   // timing-only ablation (tools/exp/build_exp.sh): stage and store, no link loop
            acc = w[0] + (double)(lds_b - smem);
#else
#pragma unroll
            for (int k0 = 0; k0 < KREG; k0 += 4) {
              if (k0 < wmax) {
                double xv[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                  const int k = k0 + kk;
                  const uint32_t li = (k & 1) ? (lc2[(k < KREG ? k : 0) / 2] >> 16) : (lc2[(k < KREG ? k : 0) / 2] & 0xFFFFu);
                  xv[kk] = load_fixed((const XT*)(lds_b + li), fill);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                  const int k = k0 + kk;
                  if (k < KREG) {
                    const double p = w[k] * xv[kk];
                    acc = acc + p;
                  }
                }
              }
            }
#endif
            acc = len > 0 ? acc : 0.0;
            pend_out[r] = (YT)epilogue(acc, dead);
            pend_off[r] = yw.off;
          }
          n_pend = r + 1;
          yw.next();
        }
      }
      group ^= 1;
    }
    if (row_live) flush_pending();
  } else if constexpr (R == 1) {
    const XT* lds_x = (const XT*)smem;
    u32x4 v[NP];
    auto load_row = [&](int64_t xoff) {
      const XT* __restrict__ xrow = (const XT*)a.x + xoff;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        if (k >= np_w) break;  // wave-uniform; invalid lanes read offset 0 of the row (in bounds, unused)
        const u32x4_u* src = (const u32x4_u*)(xrow + poff[k]);
        v[k] = (NT & 1) ? __builtin_nontemporal_load(src) : *src;
      }
    };
    // the row-end piece (clamped load, see above) exists in at most one block per row: keep its
    // element shuffle out of the common path with a wave-uniform branch
    const bool any_shifted = __builtin_amdgcn_readfirstlane((int)__any(shifted != 0)) != 0;
    // Every lane writes its pieces to their natural slots (tid + k*T): the LDS request is rounded up
    // to whole rounds of T pieces (tile_launch_cfg), so lanes past the tile's last piece write into
    // padding nobody reads and the writes need no per-lane predicate (wave-uniform k < np_w only).
    auto store_tile = [&]() {
      if (!any_shifted) {
        if (kFixAtStage && fill) {
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            if (k >= np_w) break;
            *(u32x4*)(smem + (size_t)(tid + k * T) * 16) = fix_piece<XT>(v[k]);
          }
        } else {
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            if (k >= np_w) break;
            *(u32x4*)(smem + (size_t)(tid + k * T) * 16) = v[k];
          }
        }
        return;
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        if (k < np_w) {
          u32x4 piece = v[k];
          if ((shifted >> k) & 1u) {  // last piece of the row: move the valid tail to the front
            XT tmp[kElemsPerPiece];
            __builtin_memcpy(tmp, &piece, 16);
            XT out[kElemsPerPiece];
#pragma unroll
            for (int e = 0; e < kElemsPerPiece; ++e) {
              XT val = (XT)0;
#pragma unroll
              for (int q = 0; q < kElemsPerPiece; ++q)
                if (q == e + shift_amt) val = tmp[q];
              out[e] = val;
            }
            __builtin_memcpy(&piece, out, 16);
          }
          *(u32x4*)(smem + (size_t)(tid + k * T) * 16) = (kFixAtStage && fill) ? fix_piece<XT>(piece) : piece;
        }
      }
    };

    RowWalker xw(j_begin, l, a.n_inner, a.xs_o, a.xs_l, a.xs_i);   // row being prefetched
    RowWalker yw(j_begin, l, a.n_inner, a.ys_o, a.ys_l, a.ys_i);   // row being written
#ifndef SMM_EXP_SKIP_STAGE
    load_row(xw.off);
#else
#pragma unroll
    for (int k = 0; k < NP; ++k) v[k] = u32x4{0, 0, 0, 0};
#endif
    // Single-wave workgroups issue the Y store of row j after the LDS writes of row j+1: the wait
    // for the prefetched pieces (in-order memory counter) then never includes the latest store's
    // acknowledgement (cfg3 -6 %).  Four-wave workgroups store at once (deferring cost cfg4s 13 %).
    constexpr bool kDeferStore = (WPB == 1);
    YT pend_out = (YT)0;
    int64_t pend_off = 0;
    auto flush_pending = [&]() {
      YT* __restrict__ yrow = (YT*)a.y + pend_off;
      if (NT & 2)
        __builtin_nontemporal_store(pend_out, yrow + dy);
      else
        yrow[dy] = pend_out;
    };
    for (int64_t j = j_begin; j < j_end; ++j) {
      // The counts are loop invariant, and hipcc would keep one 64-bit predicate per piece / link
      // group (k < np_w, k0 < wmax) live across the walk -- dozens of SGPR pairs, spilled to VGPR
      // lanes and read back in front of every piece.  Opaque copies make it compare the scalar again.
      asm volatile("" : "+s"(np_w), "+s"(wmax));
      // Likewise it would unpack the LDS offsets once, into one VGPR per link (48 instead of 24): with
      // 16 staging pieces that tips the 48-link kernels over 256 VGPRs, and the spill it chose was the
      // last two prefetched pieces -- a scratch store behind s_waitcnt vmcnt(0) right after the
      // prefetch was issued, i.e. no overlap of the link loop with the loads at all.
      if (MAXK > 16) {
#pragma unroll
        for (int q = 0; q < KREG / 2; ++q) asm volatile("" : "+v"(lc2[q]));
      }
      store_tile();
      if (kDeferStore && row_live && j > j_begin) flush_pending();
      __syncthreads();
#ifndef SMM_EXP_SKIP_STAGE
      if (j + 1 < j_end) {
        xw.next();
        load_row(xw.off);
      }
#endif
      if (slice_live) {
        double acc = 0.0;
        if constexpr (SPLIT) {
          // lane groups take their turn: group g continues the sums group g-1 handed over
          const int rows_blk = 64 >> a.sub_shift;
          for (int g = 0; g < n_grp; ++g) {
            double t = acc;
#pragma unroll
            for (int k0 = 0; k0 < KREG; k0 += 4) {
              if (k0 < wmax) {
                double xv[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                  const int k = k0 + kk;
                  const uint32_t li = (k & 1) ? (lc2[(k < KREG ? k : 0) / 2] >> 16)
                                              : (lc2[(k < KREG ? k : 0) / 2] & 0xFFFFu);
                  xv[kk] = load_fixed((const XT*)((const char*)lds_x + li), fill && !kFixAtStage);
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                  const int k = k0 + kk;
                  if (k < KREG) {
                    const double p = w[k] * xv[kk];
                    t = t + p;
                  }
                }
              }
            }
            acc = (grp == g && len > 0) ? t : acc;
            const double handed = __shfl(acc, (lane - rows_blk) & 63);
            acc = (grp == g + 1) ? handed : acc;
          }
#ifdef SMM_EXP_SKIP_COMPUTE
        } else if (true) {   // timing-only ablation: one LDS read instead of the link loop
          acc = (double)lds_x[lane];
#endif
        } else if (MAXK > 0) {
#pragma unroll
          for (int k0 = 0; k0 < KREG; k0 += 4) {
            if (k0 < wmax) {  // wave-uniform guard: whole groups of slots are skipped, indices stay static
              double xv[4];
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) {
                const int k = k0 + kk;
                const uint32_t li = (k & 1) ? (lc2[(k < KREG ? k : 0) / 2] >> 16)
                                            : (lc2[(k < KREG ? k : 0) / 2] & 0xFFFFu);
                xv[kk] = load_fixed((const XT*)((const char*)lds_x + li), fill && !kFixAtStage);  // unconditional: offset 0 for unused slots
              }
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) {
                const int k = k0 + kk;
                if (k < KREG) {
                  const double p = w[k] * xv[kk];
                  acc = acc + p;
                }
              }
            }
          }
        } else {
          // rows longer than 48 links: links streamed from L2, eight slots per round trip
          for (int k0 = 0; k0 < wmax; k0 += 8) {
            int32_t li[8];
            double wv[8], xv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const int kc = min(k0 + q, nslots - 1);
              li[q] = cp[(int64_t)kc * 64];
              wv[q] = vp[(int64_t)kc * 64];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) xv[q] = (double)lds_x[k0 + q < len ? li[q] : 0];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              const double p = wv[q] * xv[q];
              const double sum = acc + p;
              acc = (k0 + q < len) ? sum : acc;
            }
          }
        }
        if (MAXK > 0 && !SPLIT) acc = len > 0 ? acc : 0.0;   // a row without links never looks at the tile
        pend_out = (YT)epilogue(acc, dead);
        pend_off = yw.off;
        if (!kDeferStore && row_live) flush_pending();
      }
      yw.next();
      __syncthreads();
    }
    if (kDeferStore && row_live) flush_pending();
  } else {
    u32x4 v[R][NP];
    auto load_row = [&](int r, int64_t xoff) {
      const XT* __restrict__ xrow = (const XT*)a.x + xoff;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        if (k >= np_w) break;  // wave-uniform; invalid lanes read offset 0 of the row (in bounds, unused)
        const u32x4_u* src = (const u32x4_u*)(xrow + poff[k]);
        v[r][k] = (NT & 1) ? __builtin_nontemporal_load(src) : *src;
      }
    };
    // the row-end piece (clamped load, see above) exists in at most one block per row: keep its
    // element shuffle out of the common path with a wave-uniform branch
    const bool any_shifted = __builtin_amdgcn_readfirstlane((int)__any(shifted != 0)) != 0;
    const int tile_bytes = R > 1 ? a.tile_bytes : 0;  // LDS region of batch row r of a step: r * tile_bytes
    auto store_tile = [&](int r) {   // unconditional natural-slot writes, fill on the way in (see R == 1)
      char* region = smem + r * tile_bytes;
      if (!any_shifted) {
        if (kFixAtStage && fill) {
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            if (k >= np_w) break;
            *(u32x4*)(region + (size_t)(tid + k * T) * 16) = fix_piece<XT>(v[r][k]);
          }
        } else {
#pragma unroll
          for (int k = 0; k < NP; ++k) {
            if (k >= np_w) break;
            *(u32x4*)(region + (size_t)(tid + k * T) * 16) = v[r][k];
          }
        }
        return;
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        if (k < np_w) {
          u32x4 piece = v[r][k];
          if ((shifted >> k) & 1u) {  // last piece of the row: move the valid tail to the front
            XT tmp[kElemsPerPiece];
            __builtin_memcpy(tmp, &piece, 16);
            XT out[kElemsPerPiece];
#pragma unroll
            for (int e = 0; e < kElemsPerPiece; ++e) {
              XT val = (XT)0;
#pragma unroll
              for (int q = 0; q < kElemsPerPiece; ++q)
                if (q == e + shift_amt) val = tmp[q];
              out[e] = val;
            }
            __builtin_memcpy(&piece, out, 16);
          }
          *(u32x4*)(region + (size_t)(tid + k * T) * 16) = (kFixAtStage && fill) ? fix_piece<XT>(piece) : piece;
        }
      }
    };

    // xw.off / yw.off: the next batch row to load / to write.  A step covers rows jb .. jb+R-1; rows
    // past the walk's end repeat its last row (loaded again, never stored).
    RowWalker xw(j_begin, l, a.n_inner, a.xs_o, a.xs_l, a.xs_i);
    RowWalker yw(j_begin, l, a.n_inner, a.ys_o, a.ys_l, a.ys_i);
    auto load_step = [&](int64_t jb) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        load_row(r, xw.off);
        if (jb + r + 1 < j_end) xw.next();
      }
    };
#ifndef SMM_EXP_SKIP_STAGE
    load_step(j_begin);
#else
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int k = 0; k < NP; ++k) v[r][k] = u32x4{0, 0, 0, 0};
#endif
    // Single-wave workgroups issue the Y store of row j after the LDS writes of row j+1: the wait
    // for the prefetched pieces (in-order memory counter) then never includes the latest store's
    // acknowledgement (cfg3 -6 %).  Four-wave workgroups store at once (deferring cost cfg4s 13 %).
    constexpr bool kDeferStore = (WPB == 1 && R == 1);
    YT pend_out = (YT)0;
    int64_t pend_off = 0;
    auto flush_pending = [&]() {
      YT* __restrict__ yrow = (YT*)a.y + pend_off;
      if (NT & 2)
        __builtin_nontemporal_store(pend_out, yrow + dy);
      else
        yrow[dy] = pend_out;
    };
    for (int64_t jb = j_begin; jb < j_end; jb += R) {
      asm volatile("" : "+s"(np_w), "+s"(wmax));   // see the R == 1 walk
#pragma unroll
      for (int r = 0; r < R; ++r) store_tile(r);
      if (kDeferStore && row_live && jb > j_begin) flush_pending();
      __syncthreads();
#ifndef SMM_EXP_SKIP_STAGE
      if (jb + R < j_end) load_step(jb + R);
#endif
      if (slice_live) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const XT* lds_r = (const XT*)(smem + r * tile_bytes);
          double acc = 0.0;
#ifdef SMM_EXP_SKIP_COMPUTE
          if (true) {   // timing-only ablation: one LDS read instead of the link loop
            acc = (double)lds_r[lane];
          } else
#endif
          if (MAXK > 0) {
#pragma unroll
            for (int k0 = 0; k0 < KREG; k0 += 4) {
              if (k0 < wmax) {  // wave-uniform guard: whole groups of slots are skipped, indices stay static
                double xv[4];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                  const int k = k0 + kk;
                  const uint32_t li = (k & 1) ? (lc2[(k < KREG ? k : 0) / 2] >> 16)
                                              : (lc2[(k < KREG ? k : 0) / 2] & 0xFFFFu);
                  xv[kk] = load_fixed((const XT*)((const char*)lds_r + li), fill && !kFixAtStage);  // unconditional: offset 0 for unused slots
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                  const int k = k0 + kk;
                  if (k < KREG) {
                    const double p = w[k] * xv[kk];
                    acc = acc + p;
                  }
                }
              }
            }
          } else {
#pragma unroll 4
            for (int k = 0; k < wmax; ++k) {
              const int kc = min(k, nslots - 1);
              const bool on = k < len;
              const int32_t li = cp[(int64_t)kc * 64];
              const double xv = load_fixed(lds_r + (on ? li : 0), fill && !kFixAtStage);
              const double p = vp[(int64_t)kc * 64] * xv;
              const double sum = acc + p;
              acc = on ? sum : acc;
            }
          }
          if (MAXK > 0) acc = len > 0 ? acc : 0.0;   // a row without links never looks at the tile
          pend_out = (YT)epilogue(acc, dead);
          pend_off = yw.off;
          if (!kDeferStore && row_live && jb + r < j_end) flush_pending();
          if (jb + r + 1 < j_end) yw.next();
        }
      }
      __syncthreads();
    }
    if (kDeferStore && row_live) flush_pending();
  }
}

// ------------------------------------------------------------------ kernel C
// Batch-fastest operand layout ("SB"): X is (S, B) -- the B batch values of one source cell are
// contiguous -- and Y is still written (B, D) as the reference lays it out (regrid.py:550).  In
// the native (B, S) layout a bilinear 4:1 stencil needs 16-B pairs on a 32-B stride, so half of
// every fetched 128-B line is wasted (element-granular fraction capped at ~0.55).  With
// the batch fastest every needed source cell is a contiguous run: fetched bytes == algorithmic
// bytes, and every link is one coalesced 1-KiB wave load (64 lanes x 16 B).
//
// One wave owns a tile of TD destination rows x BT = 64 * VEC batch entries.  Links are wave
// uniform: column / weight come through scalar loads straight from the canonical CSR (no SELL
// padding, ragged rows cost nothing), the tile's links are walked as one flat sequence in batches
// of U loads (two batches in flight), and a row's sum is finished -- fill on load, epilogue on
// flush -- when the walk crosses its end.  Accumulation is per row in ascending source order with
// separate multiply and add: bit-identical to the oracle.  Finished rows go to an LDS tile
// [TD][BT] that is read back transposed, so that Y is written in runs of TD consecutive
// destination cells per batch row (TD = 16 doubles = one 128-B line).

// YSB: the result is kept batch-fastest too -- Y (D, ldy >= B), the layout a second regrid consumes
// without a transpose (SMM_APPLY_SB_Y_SB).  The finished rows still wait in the LDS tile and leave at
// the end of the tile, each as one contiguous 1-KiB wave store: a store issued in the middle of the
// link walk sits in the in-order memory counter in front of the loads that follow it, and the walk
// then waits for its acknowledgement (measured: config 2 1.70 ms with stores in the walk).
// (Workgroups of 2 / 4 waves sharing one tile -- 18 / 20 instead of 9 waves per CU -- were built and
// measured level with single-wave workgroups, 1.69 / 1.59 vs 1.62 ms on config 2: the kernel runs at
// the rate the chip sustains for this read : write mix, occupancy is not what bounds it.)
// The CSR and epilogue vectors of one operator as the kernel body reads them: plain kernel-argument pointers
// (single operator) or constant-address-space views (level group, see SbGroupArgs).
template <typename I64P, typename I32P, typename F64P, typename U8P>
struct SbMatrix {
  I64P rowptr;
  I32P col;
  F64P val;
  U8P imask;
  F64P frac;
};
struct SbTile {   // what does not depend on the level
  const void* x;
  void* y;
  int64_t ldx, ldy, n_batch, n_dst, n_dtiles, n_btiles;
  uint32_t n_blocks;   // blocks of this operator's grid (for the XCD remap)
  double area_min;
  int masked, xcd_remap, b_fastest;
};

template <typename XT, typename YT, int TD, int U, bool FILL, bool YSB, typename M>
__device__ __forceinline__ void sb_tile_body(const M& m, const SbTile& a, uint32_t bid) {
  constexpr int VEC = 2;                    // batch entries per lane
  constexpr int BT = 64 * VEC;              // batch entries per tile
  constexpr int PAD = 16 / (int)sizeof(YT); // LDS row padding: one 16-B slot (conflict-free transposed reads)
  constexpr int LROW = BT + PAD;
  static_assert(TD % 2 == 0 && 128 % TD == 0 && TD <= 64, "store phase: TD / 2 lanes per batch row");
  __shared__ __attribute__((aligned(16))) YT tile[TD * LROW];
  typedef XT xvec __attribute__((ext_vector_type(VEC)));
  typedef xvec xvec_u __attribute__((aligned(sizeof(XT))));   // element-aligned (any ldx / base)

  const int lane = threadIdx.x;
  // grids stay below 2^31 blocks: 32-bit index arithmetic
  if (a.xcd_remap > 0) {   // runs of consecutive tiles (neighbours in space) share one XCD's L2
    const uint32_t C = (uint32_t)a.xcd_remap, round = 8 * C;
    if (bid < (a.n_blocks / round) * round) {
      const uint32_t xcd = bid & 7, slot = bid >> 3;
      bid = (slot / C) * round + xcd * C + (slot % C);
    }
  }
  int64_t dt, bt;
  if (a.b_fastest > 0) {
    // strips of `b_fastest` destination tiles: inside a strip the destination tile runs fastest, then the
    // batch tile -- consecutive workgroups (one XCD run) read consecutive 1-KiB pieces of the same source
    // rows (b) and write neighbouring 128-B lines of the same Y rows (d)
    const uint32_t sub = (uint32_t)a.b_fastest, per = (uint32_t)a.n_btiles * sub;
    const uint32_t sd = bid / per, rem = bid - sd * per;
    const uint32_t left = (uint32_t)a.n_dtiles - sd * sub, sub_here = left < sub ? left : sub;
    const uint32_t b32 = rem / sub_here;
    bt = b32;
    dt = (int64_t)sd * sub + (rem - b32 * sub_here);
  } else {
    const uint32_t bt32 = bid / (uint32_t)a.n_dtiles;
    dt = bid - bt32 * (uint32_t)a.n_dtiles;
    bt = bt32;
  }
  // (Round 6: dealing the tiles of two neighbouring rows of the destination grid to consecutive workgroups -- so that
  // one XCD's L2 sees both users of the source cells along their common edge -- bought 0.5 - 4.6 % on config 3 kept
  // batch-fastest, below the 5 % bar: profiles/r06_cfg3sb_lat_pair_order.txt; not kept.)
  const int64_t d0 = dt * TD;
  const int rows = (int)(a.n_dst - d0 < TD ? a.n_dst - d0 : TD);
  const int64_t b0 = bt * BT;
  // lanes past the batch end load the last valid pair (in bounds) and store nothing
  int64_t bl = b0 + (int64_t)lane * VEC;
  const int64_t b_last = a.n_batch >= VEC ? a.n_batch - VEC : 0;
  const bool tiny_batch = a.n_batch < VEC;
  // odd batch size: the lane holding the last entry loads the pair one element earlier and takes
  // its second element (lanes further out have no valid entry at all)
  const bool shift1 = !tiny_batch && bl == a.n_batch - 1;
  if (bl > b_last) bl = b_last;
  const XT* __restrict__ xl = (const XT*)a.x + bl;

  const int64_t p0 = m.rowptr[d0], p1 = m.rowptr[d0 + rows];
  int d_local = 0;
  int64_t row_end = m.rowptr[d0 + 1];
  int64_t row_end_next = m.rowptr[d0 + (rows > 1 ? 2 : 1)];   // scalar prefetch, one row ahead
  double acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.0;

  // regrid.py:553-565 per destination row, gathered once per tile (lane r looks at row r) into a
  // wave-uniform bit mask: a vector load inside flush_row would make every row end wait for all
  // outstanding X loads
  bool dead_lane = false;
  if (lane < rows) {
    if (a.masked && m.imask) dead_lane = m.imask[d0 + lane] == 0;
    if (a.area_min > 0.0 && m.frac) dead_lane = dead_lane || (m.frac[d0 + lane] < a.area_min);
  }
  const unsigned long long dead_mask = __ballot(dead_lane);

  auto flush_row = [&]() {
    const bool dead = (dead_mask >> d_local) & 1ull;   // wave-uniform
    YT out[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      out[v] = (YT)epilogue(acc[v], dead);
      acc[v] = 0.0;
    }
    typedef YT yvec __attribute__((ext_vector_type(VEC)));
    yvec o;
#pragma unroll
    for (int v = 0; v < VEC; ++v) o[v] = out[v];
    *(yvec*)(&tile[d_local * LROW + lane * VEC]) = o;
    ++d_local;
    row_end = row_end_next;
    const int nxt = d_local + 2 <= rows ? d_local + 2 : rows;
    row_end_next = m.rowptr[d0 + nxt];
  };

  xvec xv[2][U];
  double w[2][U];
  auto load_batch = [&](int buf, int64_t base) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t p = base + u;
      if (p > p1 - 1) p = p1 - 1;            // padding repeats the tile's last link (valid address)
      const int64_t c = m.col[p];
      w[buf][u] = m.val[p];
      xv[buf][u] = *(const xvec_u*)(xl + c * a.ldx);
    }
  };
  auto consume = [&](int buf, int64_t base) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t p = base + u;
      if (p < p1) {                           // wave-uniform
        while (row_end <= p) flush_row();     // rows ending before this link (empty rows included)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          XT e = (v == 0 && shift1) ? xv[buf][u][1] : xv[buf][u][v];
          if (FILL) e = __builtin_isfinite(e) ? e : (XT)1e20;   // regrid.py:545-547, dtype's own 1e20
          const double prod = w[buf][u] * (double)e;
          acc[v] = acc[v] + prod;
        }
      }
    }
  };

  if (p1 > p0 && !tiny_batch) {
    load_batch(0, p0);
    for (int64_t base = p0; base < p1; base += 2 * U) {
      if (base + U < p1) load_batch(1, base + U);
      consume(0, base);
      if (base + 2 * U < p1) load_batch(0, base + 2 * U);
      if (base + U < p1) consume(1, base + U);
    }
  } else if (p1 > p0) {
    // fewer batch entries than one lane's vector: element-wise walk (never on a hot path)
    for (int64_t p = p0; p < p1; ++p) {
      while (row_end <= p) flush_row();
      const int64_t c = m.col[p];
      const double wv = m.val[p];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int64_t b = b0 + (int64_t)lane * VEC + v;
        XT e = ((const XT*)a.x)[c * a.ldx + (b < a.n_batch ? b : a.n_batch - 1)];
        if (FILL) e = __builtin_isfinite(e) ? e : (XT)1e20;
        const double prod = wv * (double)e;
        acc[v] = acc[v] + prod;
      }
    }
  }
  while (d_local < rows) flush_row();
  __syncthreads();

  if constexpr (YSB) {
    // batch-fastest result: row r's 128 batch entries are one contiguous run of Y row d0 + r
    typedef YT yvec __attribute__((ext_vector_type(VEC)));
    typedef yvec yvec_u __attribute__((aligned(sizeof(YT))));
    const int64_t b = b0 + (int64_t)lane * VEC;
    YT* __restrict__ yb = (YT*)a.y + d0 * a.ldy + b;
#pragma unroll 4
    for (int r = 0; r < rows; ++r) {
      const yvec o = *(const yvec*)(&tile[r * LROW + lane * VEC]);
      YT* dst = yb + (int64_t)r * a.ldy;
      if (b + VEC <= a.n_batch)
        __builtin_nontemporal_store(o, (yvec_u*)dst);
      else if (b < a.n_batch)
        __builtin_nontemporal_store(o[0], dst);   // odd batch: the last entry (shift1 put it in element 0)
    }
    return;
  }

  // transposed read-back: TD / 2 lanes cover one batch row's TD destination cells (two per lane)
  constexpr int LPB = TD / 2;            // lanes per batch row
  constexpr int BPI = 64 / LPB;          // batch rows per store instruction
  const int dp = lane % LPB, bsub = lane / LPB;
  const int64_t dd = d0 + 2 * dp;
  YT* __restrict__ yb = (YT*)a.y + dd;
#pragma unroll 4
  for (int i = 0; i < BT / BPI; ++i) {
    const int bloc = i * BPI + bsub;
    const int64_t b = b0 + bloc;
    const YT v0 = tile[(2 * dp) * LROW + bloc];
    const YT v1 = tile[(2 * dp + 1) * LROW + bloc];
    if (b < a.n_batch) {
      YT* dst = yb + b * a.ldy;
      if (dd + 1 < a.n_dst) {
        typedef YT y2 __attribute__((ext_vector_type(2)));
        typedef y2 y2_u __attribute__((aligned(sizeof(YT))));
        y2 o;
        o[0] = v0;
        o[1] = v1;
        __builtin_nontemporal_store(o, (y2_u*)dst);
      } else if (dd < a.n_dst) {
        __builtin_nontemporal_store(v0, dst);
      }
    }
  }
}

template <typename XT, typename YT, int TD, int U, bool FILL, bool YSB = false>
__global__ __launch_bounds__(64) void smm_apply_sb_kernel(SbArgs a) {
  const SbMatrix<const int64_t*, const int32_t*, const double*, const uint8_t*> m{a.rowptr, a.col, a.val, a.imask, a.frac};
  const SbTile t{a.x, a.y, a.ldx, a.ldy, a.n_batch, a.n_dst, a.n_dtiles, a.n_btiles, (uint32_t)a.n_blocks,
                 a.area_min, a.masked, a.xcd_remap, a.b_fastest};
  sb_tile_body<XT, YT, TD, U, FILL, YSB>(m, t, blockIdx.x);
}

// The same tiles for every data level of a group in one launch: workgroup -> (level, tile of that level's grid).
// Levels are independent, so the dispatcher backfills the thin deep levels' tails with the next level's tiles -- no
// ramp-up and tail per level as with one launch each.
template <typename XT, typename YT, int TD, int U, bool FILL, bool YSB = false>
__global__ __launch_bounds__(64) void smm_group_apply_sb_kernel(SbGroupArgs a) {
  typedef const __attribute__((address_space(4))) int64_t* k_i64;
  typedef const __attribute__((address_space(4))) int32_t* k_i32;
  typedef const __attribute__((address_space(4))) double* k_f64;
  typedef const __attribute__((address_space(4))) uint8_t* k_u8;
  const uint32_t per = (uint32_t)a.blocks_per_level;
  const uint32_t lvl = blockIdx.x / per;          // wave-uniform: the level's pointers come through scalar loads
  const uint32_t bid = blockIdx.x - lvl * per;
  const SbLevelPtrs L = a.lev[lvl];
  // the operator's arrays are immutable while a kernel runs: constant address space keeps their loads scalar
  const SbMatrix<k_i64, k_i32, k_f64, k_u8> m{(k_i64)L.rowptr, (k_i32)L.col, (k_f64)L.val, (k_u8)L.imask, (k_f64)L.frac};
  const SbTile t{(const XT*)a.x + (int64_t)lvl * a.xs_lev, (YT*)a.y + (int64_t)lvl * a.ys_lev, a.ldx, a.ldy, a.n_batch,
                 a.n_dst, a.n_dtiles, a.n_btiles, per, a.area_min, L.imask != nullptr, a.xcd_remap, a.b_fastest};
  sb_tile_body<XT, YT, TD, U, FILL, YSB>(m, t, bid);
}

// counter-based synthetic field: splitmix64 -> two uniforms -> Box-Muller
template <typename T>
__global__ void smm_fill_random_kernel(T* __restrict__ dst, int64_t n, uint64_t seed, double mean,
                                       double sigma) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const float u1 = ((float)((z >> 40) + 1)) * (1.0f / 16777217.0f);      // (0, 1)
    const float u2 = (float)((z >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);  // [0, 1)
    const float r = sqrtf(-2.0f * __logf(u1));
    const float g = r * __cosf(6.28318530718f * u2);
    dst[i] = (T)(mean + sigma * (double)g);
  }
}

// dst_imask[d] = y[d] < 0.5 ? 0 : 1      (weights.py:51)
__global__ void smm_mask_threshold_kernel(const double* __restrict__ y, int32_t* __restrict__ m,
                                          int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) m[i] = (y[i] < 0.5) ? 0 : 1;
}

}  // namespace
