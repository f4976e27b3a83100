// Kernel launch templates, one explicit instantiation per (X dtype, Y dtype) pair
// (smm_launch_inst.hip is compiled four times, in parallel): the tile kernel alone has several
// hundred instantiations, which one translation unit would compile for minutes.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <type_traits>

#include "../../include/smmregrid_amd.h"
#include "smm_kernels.hpp"

#pragma clang fp contract(off)

namespace smm {
int fail_msg(int code, const std::string& msg);   // sets the thread-local error text (smm_device.hip)
int tuning(int knob);                             // smm_debug_set_tuning's current value (0 = library default)
}

namespace smm_launch {

#define SMM_LAUNCH_HIP(call)                                                              \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      (void)hipGetLastError();                                                            \
      return smm::fail_msg(SMM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    }                                                                                     \
  } while (0)

// batch rows per thread of the SELL kernel for a batch of n_j rows
inline int sell_batch_rows(int64_t n_j) {
  const int want = smm::tuning(SMM_TUNE_SELL_BATCH_ROWS);
  if (n_j >= 8 && want == 8) return 8;
  if (n_j >= 4 && want != 2) return 4;
  return n_j >= 2 ? 2 : 1;
}

// Extra dynamic LDS requested for every wave of the batch-fastest kernels (tuning knob SMM_TUNE_SB_LDS_PAD, bytes): it
// caps the waves per CU -- the 16.6-KB tile alone allows 9.  Measured (profiles/r05_sb_experiments.txt): config 3 is level
// down to 7 waves per CU and loses below; config 2 gains 0 - 5 % at 6 - 8.  No pad by default.
inline size_t sb_lds_pad() { return (size_t)std::max(smm::tuning(SMM_TUNE_SB_LDS_PAD), 0); }

// consecutive logical blocks given to one XCD (tile and batch-fastest kernels); 0 = dispatcher order
inline int xcd_run_length() {
  const int want = smm::tuning(SMM_TUNE_XCD_RUN);
  return want < 0 ? 0 : (want > 0 ? want : 32);
}

template <typename XT, typename YT>
int launch_sell(const ApplyArgs& a, int64_t n_lev, bool fill, unsigned flags, hipStream_t s) {
  ApplyArgs args = a;
  auto go = [&](auto bt_tag) -> int {
    constexpr int BT = decltype(bt_tag)::value;
    args.n_jtiles = (a.n_j + BT - 1) / BT;
    const int64_t total = args.n_dblocks * args.n_jtiles * n_lev;
    if (total <= 0) return SMM_OK;
    if (total > 0x7fffffffLL) return smm::fail_msg(SMM_ERR_INVALID, "launch grid exceeds 2^31-1 blocks");
    hipLaunchKernelGGL((smm_apply_sell_kernel<XT, YT, BT>), dim3((unsigned)total), dim3(kThreads), 0,
                       s, args, fill);
    SMM_LAUNCH_HIP(hipGetLastError());
    return SMM_OK;
  };
  // batch rows per thread: 4 (gather-bound operators want many waves in flight: scatter -3 %,
  // config 2 -1 % against 8); SMM_TUNE_SELL_BATCH_ROWS asks for 8 or 2
  switch (sell_batch_rows(a.n_j)) {
    case 8: return go(std::integral_constant<int, 8>());
    case 4: return go(std::integral_constant<int, 4>());
    case 2: return go(std::integral_constant<int, 2>());
    default: return go(std::integral_constant<int, 1>());
  }
}

// Launch geometry of the tile kernel, shared by the launcher and smm_operator_launch_info.
struct TileLaunchCfg {
  int j_per_block = 0;     // batch rows walked by one workgroup
  int64_t n_jtiles = 0;
  int64_t total = 0;       // workgroups
  int xcd_remap = 0;
  int threads = 0;
  int np_needed = 0;       // 16-B staging pieces per thread of the widest block
  int rows = 1;            // batch rows staged per barrier pair (R)
  bool dma = false;        // staging by LDS-DMA into a ring of two tile slots (single-row steps)
  size_t tile = 0, lds = 0;
  bool big_operator = false;
};

inline TileLaunchCfg tile_launch_cfg(const ApplyArgs& a, int64_t n_lev, int tile_which, int64_t max_chunks,
                                     int64_t max_row_nnz, unsigned flags, size_t xsz) {
  TileLaunchCfg c;
  (void)flags;
  const int jpb = std::max(smm::tuning(SMM_TUNE_TILE_WALK), 0);
  const int staging = smm::tuning(SMM_TUNE_TILE_STAGING);          // 1 = registers, 2 = LDS-DMA forced
  const int rows_knob = smm::tuning(SMM_TUNE_TILE_ROWS_PER_STEP);  // 1 / 2 / 4
  // Batch rows walked per workgroup: the prologue (links -> registers) is amortised over
  // them, so heavier rows want longer walks; keep >= ~4096 workgroups to fill 256 CUs.
  // An operator that does not stay in L2 (links x 12 B beyond ~32 MB) is re-read from HBM by every
  // walk: amortise it over long walks whatever the row length.
  c.big_operator = a.n_dst * std::max<int64_t>(max_row_nnz, 1) * 12 > (32ll << 20);
  int64_t walk = jpb ? (int64_t)jpb : (c.big_operator ? 128 : (max_row_nnz <= 4 ? 4 : 64));
  if (!jpb)
    while (walk > 1 && a.n_dblocks * ((a.n_j + walk - 1) / walk) * n_lev < 4096) walk /= 2;
  c.j_per_block = (int)std::min<int64_t>(a.n_j, walk);
  c.n_jtiles = c.j_per_block > 0 ? (a.n_j + c.j_per_block - 1) / c.j_per_block : 0;
  c.total = a.n_dblocks * c.n_jtiles * n_lev;
  // runs of 32 consecutive blocks per XCD (SMM_TUNE_XCD_RUN: another run length, -1 = dispatcher order)
  c.xcd_remap = xcd_run_length();
  const int64_t max_pieces = max_chunks * (int64_t)(kChunkElems * xsz / 16);
  c.threads = (tile_which ? 1 : kWavesPerBlock) * 64;  // == tile_waves(MAXK) * 64
  c.np_needed = (int)((max_pieces + c.threads - 1) / c.threads);
  // whole rounds of one 16-B piece per thread: every lane stores its pieces unpredicated, the
  // last round's surplus lands in padding
  c.tile = (size_t)std::max(c.np_needed, 1) * c.threads * 16;
  // Small tiles (one or two 16-B pieces per thread, 4-wave shape): a step of 4 / 2 batch rows per
  // barrier pair keeps as many bytes in flight as a full tile would (SMM_TUNE_TILE_ROWS_PER_STEP lowers it).
  c.rows = 1;
  if (!tile_which) c.rows = c.np_needed <= 1 ? 4 : (c.np_needed <= 2 ? 2 : 1);
  while (c.rows > 1 && (c.rows > c.j_per_block || (rows_knob > 0 && c.rows > rows_knob))) c.rows /= 2;
  // LDS-DMA staging instead (smm_kernels.hpp, DMA): on by default for the small tiles the multi-row
  // steps serve (two slots of <= 8 KB keep every workgroup slot of the CU), SMM_TUNE_TILE_STAGING = 2 forces it
  // for any tile of either block shape that keeps its links in registers, 1 switches it off.
  // The DMA moves aligned 16-B pieces: base, strides and row length must be multiples of 16 B.
  const bool aligned = ((uintptr_t)a.x % 16 == 0) && (a.xs_o * (int64_t)xsz % 16 == 0) &&
                       (a.xs_l * (int64_t)xsz % 16 == 0) && (a.xs_i * (int64_t)xsz % 16 == 0) &&
                       (a.n_src * (int64_t)xsz % 16 == 0);
  const bool wanted = staging == 2 || (staging == 0 && !tile_which && c.np_needed <= 2);
  c.dma = wanted && aligned && a.sub_shift == 0 && max_row_nnz > 0 && max_row_nnz <= 48 && 2 * c.tile <= 65536;
  if (c.dma) {
    // Rows per step of the DMA ring (two groups of `rows` slots).  Tiles of one piece per thread (4 KB:
    // coarse -> fine regrids, whose steps are pure round trips) take two rows per step -- 16 KB per
    // workgroup still keeps every workgroup slot of the CU (upsampling r360x180 -> r1440x721 2.71 -> 2.39 ms);
    // with 8-KB tiles two rows cost workgroups (config-4 geometry 4.71 -> 5.05 ms, four rows 5.92).
    // SMM_TUNE_TILE_ROWS_PER_STEP picks one, two or four.
    c.rows = 1;
    if (!tile_which && c.np_needed <= 4)
      c.rows = (rows_knob == 1 || rows_knob == 2 || rows_knob == 4) ? rows_knob : (c.np_needed <= 1 ? 2 : 1);
    while (c.rows > 1 && (c.rows > c.j_per_block || 2 * c.rows * c.tile > 65536)) c.rows /= 2;
  }
  c.lds = c.dma ? 2 * (size_t)c.rows * c.tile : c.tile * (size_t)c.rows;
  return c;
}

template <typename XT, typename YT>
int launch_tile(const ApplyArgs& a, int64_t n_lev, int tile_which, int64_t max_chunks,
                int64_t max_row_nnz, int tile_flags, bool fill, unsigned flags, hipStream_t s) {
  ApplyArgs args = a;
  const TileLaunchCfg cfg = tile_launch_cfg(a, n_lev, tile_which, max_chunks, max_row_nnz, flags, sizeof(XT));
  args.j_per_block = cfg.j_per_block;
  args.n_jtiles = cfg.n_jtiles;
  const int64_t total = cfg.total;
  if (total <= 0) return SMM_OK;
  if (total > 0x7fffffffLL) return smm::fail_msg(SMM_ERR_INVALID, "launch grid exceeds 2^31-1 blocks");
  args.n_blocks = total;
  args.xcd_remap = cfg.xcd_remap;
  const int np_needed = cfg.np_needed;
  const int rows = cfg.rows;
  args.tile_bytes = (int)cfg.tile;
  const size_t lds = cfg.lds;

  const bool dma = cfg.dma;   // LDS-DMA staging into a ring of two tile slots (tile_launch_cfg)
  auto go_dma = [&](auto k_tag, auto np_tag, auto nt_tag) -> int {
    constexpr int MAXK = decltype(k_tag)::value;
    if constexpr (MAXK > 0) {
      constexpr int NPV = decltype(np_tag)::value;
      constexpr int NTV = decltype(nt_tag)::value;
      const dim3 grid((unsigned)total), block(tile_waves(MAXK) * 64);
      if constexpr (MAXK <= 16 && NPV <= 4) {
        if (rows == 4) {
          hipLaunchKernelGGL((smm_apply_tile2_kernel<XT, YT, MAXK, NPV, NTV, 4, false, true>), grid, block, lds, s, args, fill);
          SMM_LAUNCH_HIP(hipGetLastError());
          return SMM_OK;
        }
        if (rows == 2) {
          hipLaunchKernelGGL((smm_apply_tile2_kernel<XT, YT, MAXK, NPV, NTV, 2, false, true>), grid, block, lds, s, args, fill);
          SMM_LAUNCH_HIP(hipGetLastError());
          return SMM_OK;
        }
      }
      hipLaunchKernelGGL((smm_apply_tile2_kernel<XT, YT, MAXK, NPV, NTV, 1, false, true>), grid, block, lds, s, args, fill);
      SMM_LAUNCH_HIP(hipGetLastError());
      return SMM_OK;
    } else {
      return smm::fail_msg(SMM_ERR_UNSUPPORTED, "LDS-DMA staging needs the links in registers");
    }
  };
  auto go3 = [&](auto k_tag, auto np_tag, auto nt_tag, auto r_tag) -> int {
    constexpr int MAXK = decltype(k_tag)::value;
    constexpr int NP = decltype(np_tag)::value;
    constexpr int NT = decltype(nt_tag)::value;
    constexpr int R = decltype(r_tag)::value;
    hipLaunchKernelGGL((smm_apply_tile2_kernel<XT, YT, MAXK, NP, NT, R>), dim3((unsigned)total),
                       dim3(tile_waves(MAXK) * 64), lds, s, args, fill);
    SMM_LAUNCH_HIP(hipGetLastError());
    return SMM_OK;
  };
  // Part-of-a-slice blocks: the idle lanes take over parts of the rows (SPLIT kernels) when a lane
  // group's share fits the link registers (SMM_TUNE_TILE_SPLIT_ROWS = 1: off, for A/B runs).
  const int n_grp = 1 << args.sub_shift;
  const bool split = args.sub_shift > 0 && smm::tuning(SMM_TUNE_TILE_SPLIT_ROWS) != 1 && max_row_nnz <= (int64_t)n_grp * 48;
  const int64_t per_grp = (max_row_nnz + n_grp - 1) / n_grp;
  auto go2 = [&](auto k_tag, auto np_tag, auto nt_tag) -> int {
    constexpr int MAXK = decltype(k_tag)::value;
    if (dma) return go_dma(k_tag, np_tag, nt_tag);
    if constexpr (MAXK == 32 || MAXK == 48) {
      if (split) {
        hipLaunchKernelGGL((smm_apply_tile2_kernel<XT, YT, MAXK, decltype(np_tag)::value, decltype(nt_tag)::value, 1, true>),
                           dim3((unsigned)total), dim3(64), lds, s, args, fill);
        SMM_LAUNCH_HIP(hipGetLastError());
        return SMM_OK;
      }
    }
    if constexpr (MAXK > 0 && MAXK <= 16) {
      if (rows == 4) return go3(k_tag, std::integral_constant<int, 1>(), nt_tag, std::integral_constant<int, 4>());
      if (rows == 2) return go3(k_tag, std::integral_constant<int, 2>(), nt_tag, std::integral_constant<int, 2>());
    }
    return go3(k_tag, np_tag, nt_tag, std::integral_constant<int, 1>());
  };
  auto with_k = [&](auto fn) -> int {  // plan shape 0 <-> 4 waves (rows of <= 16 links), shape 1.. <-> 1 wave
    if (!tile_which) {
      if (max_row_nnz > 16)
        return smm::fail_msg(SMM_ERR_UNSUPPORTED, "4-wave tile blocks serve rows of at most 16 links");
      if (max_row_nnz <= 4) return fn(std::integral_constant<int, 4>());
      if (max_row_nnz <= 8) return fn(std::integral_constant<int, 8>());
      return fn(std::integral_constant<int, 16>());
    }
    if (smm::tuning(SMM_TUNE_TILE_LINKS) == 1) return fn(std::integral_constant<int, 0>());  // tuning: stream the links
    if (split) return per_grp <= 32 ? fn(std::integral_constant<int, 32>()) : fn(std::integral_constant<int, 48>());
    if (max_row_nnz <= 32) return fn(std::integral_constant<int, 32>());
    if (max_row_nnz <= 48) return fn(std::integral_constant<int, 48>());
    return fn(std::integral_constant<int, 0>());  // longer rows: links streamed from L2
  };
  if (np_needed > 16)
    return smm::fail_msg(SMM_ERR_UNSUPPORTED, "tile plan exceeds the staging register budget");
  // Y stores are always non-temporal; X loads are non-temporal only if no staged line is shared between
  // blocks (tile_reuse false).  SMM_TUNE_TILE_X_LOADS = 1 / 2 forces non-temporal / cached X loads.  (Cached Y
  // stores measured flat or worse everywhere and are no longer built.)
  const bool tile_reuse = tile_flags & 1;
  int nt = tile_reuse ? 2 : 3;
  if (smm::tuning(SMM_TUNE_TILE_X_LOADS) == 1) nt = 3;
  if (smm::tuning(SMM_TUNE_TILE_X_LOADS) == 2) nt = 2;
  return with_k([&](auto k_tag) -> int {
    auto with_nt = [&](auto np_tag) -> int {
      if (nt == 2) return go2(k_tag, np_tag, std::integral_constant<int, 2>());
      return go2(k_tag, np_tag, std::integral_constant<int, 3>());
    };
    if (np_needed <= 4) return with_nt(std::integral_constant<int, 4>());
    if (np_needed <= 8) return with_nt(std::integral_constant<int, 8>());
    return with_nt(std::integral_constant<int, 16>());
  });
}

// Batch-fastest layout (kernel C).  TD destination rows per tile: 16 doubles = one 128-B line of Y
// per batch row; f32 output takes 32 rows for the same line.
template <typename XT, typename YT>
int launch_sb(const SbArgs& a, bool fill, unsigned flags, hipStream_t s) {
  SbArgs args = a;
  constexpr int TD = sizeof(YT) == 8 ? 16 : 32;
  constexpr int BT = 128;
  args.n_dtiles = (a.n_dst + TD - 1) / TD;
  args.n_btiles = (a.n_batch + BT - 1) / BT;
  const int64_t total = args.n_dtiles * args.n_btiles;
  if (total <= 0) return SMM_OK;
  if (total > 0x7fffffffLL) return smm::fail_msg(SMM_ERR_INVALID, "launch grid exceeds 2^31-1 blocks");
  args.n_blocks = total;
  args.xcd_remap = xcd_run_length();
  // Tile order: strips of 2 destination tiles, inside a strip destination tile fastest, then batch tile
  // (config 2: 1.76 -> 1.65 ms against destination-tile-fastest order over the whole grid; strips of 1, 4,
  // 8, 16 tiles within 3 % of it).  SMM_TUNE_SB_STRIP: -1 = whole-grid order, n = strips of n tiles.
  const int strip = smm::tuning(SMM_TUNE_SB_STRIP);
  args.b_fastest = strip < 0 ? 0 : (strip > 0 ? strip : 2);
  const bool ysb = (flags & SMM_APPLY_SB_Y_SB) != 0;   // result kept batch-fastest: Y (D, ldy >= B)
  auto go = [&](auto u_tag, auto fill_tag) {
    constexpr int UU = decltype(u_tag)::value;
    constexpr bool FF = decltype(fill_tag)::value;
    if (ysb)
      hipLaunchKernelGGL((smm_apply_sb_kernel<XT, YT, TD, UU, FF, true>), dim3((unsigned)total), dim3(64), sb_lds_pad(), s, args);
    else
      hipLaunchKernelGGL((smm_apply_sb_kernel<XT, YT, TD, UU, FF>), dim3((unsigned)total), dim3(64), sb_lds_pad(), s, args);
  };
  auto with_fill = [&](auto u_tag) {
    if (fill) go(u_tag, std::true_type());
    else go(u_tag, std::false_type());
  };
  if (smm::tuning(SMM_TUNE_SB_LOADS) == 4) with_fill(std::integral_constant<int, 4>());   // tuning: 4 loads per batch instead of 8
  else with_fill(std::integral_constant<int, 8>());
  SMM_LAUNCH_HIP(hipGetLastError());
  return SMM_OK;
}

// The batch-fastest kernel over the data levels of a group in ONE launch (smm_group_apply_sb): grid = levels x
// (destination tiles x batch tiles); the caller has filled a.lev[0 .. n_lev) and the per-level strides.
template <typename XT, typename YT>
int launch_sb_group(const SbGroupArgs& a, bool fill, unsigned flags, hipStream_t s) {
  SbGroupArgs args = a;
  constexpr int TD = sizeof(YT) == 8 ? 16 : 32;
  constexpr int BT = 128;
  args.n_dtiles = (a.n_dst + TD - 1) / TD;
  args.n_btiles = (a.n_batch + BT - 1) / BT;
  args.blocks_per_level = args.n_dtiles * args.n_btiles;
  const int64_t total = args.blocks_per_level * a.n_lev;
  if (total <= 0) return SMM_OK;
  if (total > 0x7fffffffLL) return smm::fail_msg(SMM_ERR_INVALID, "launch grid exceeds 2^31-1 blocks");
  args.xcd_remap = xcd_run_length();
  const int strip = smm::tuning(SMM_TUNE_SB_STRIP);
  args.b_fastest = strip < 0 ? 0 : (strip > 0 ? strip : 2);
  const bool ysb = (flags & SMM_APPLY_SB_Y_SB) != 0;
  auto go = [&](auto u_tag, auto fill_tag) {
    constexpr int UU = decltype(u_tag)::value;
    constexpr bool FF = decltype(fill_tag)::value;
    if (ysb)
      hipLaunchKernelGGL((smm_group_apply_sb_kernel<XT, YT, TD, UU, FF, true>), dim3((unsigned)total), dim3(64), sb_lds_pad(), s, args);
    else
      hipLaunchKernelGGL((smm_group_apply_sb_kernel<XT, YT, TD, UU, FF>), dim3((unsigned)total), dim3(64), sb_lds_pad(), s, args);
  };
  auto with_fill = [&](auto u_tag) {
    if (fill) go(u_tag, std::true_type());
    else go(u_tag, std::false_type());
  };
  if (smm::tuning(SMM_TUNE_SB_LOADS) == 4) with_fill(std::integral_constant<int, 4>());
  else with_fill(std::integral_constant<int, 8>());
  SMM_LAUNCH_HIP(hipGetLastError());
  return SMM_OK;
}

}  // namespace smm_launch
