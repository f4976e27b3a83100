// Host-side operator construction (K6 of SURVEY.md section 2): SCRIP links ->
// canonical CSR -> SELL-64 device layout and the LDS source-tile plan.
//
// Replaces what the reference gets from the sparse.COO constructor
// (weights.py:31-42): 1-based addresses become 0-based, coordinates are sorted
// and duplicate coordinates are summed.
#include "smm_internal.h"

#include <algorithm>
#include <cstring>

namespace smm {

bool build_csr(int64_t n_src, int64_t n_dst, int64_t nnz, const int32_t* src1,
               const int32_t* dst1, const double* w, HostCsr& out, std::string& err) {
  if (n_src < 0 || n_dst < 0 || nnz < 0) {
    err = "negative size";
    return false;
  }
  if (n_src > INT32_MAX || n_dst > INT32_MAX || nnz > INT32_MAX) {
    err = "sizes beyond int32 addressing (SCRIP addresses are int32)";
    return false;
  }
  if (nnz > 0 && (!src1 || !dst1 || !w)) {
    err = "null link array";
    return false;
  }
  for (int64_t k = 0; k < nnz; ++k) {
    const int64_t s = (int64_t)src1[k] - 1, d = (int64_t)dst1[k] - 1;
    if (s < 0 || s >= n_src) {
      err = "src_address[" + std::to_string(k) + "]=" + std::to_string(src1[k]) +
            " outside 1.." + std::to_string(n_src);
      return false;
    }
    if (d < 0 || d >= n_dst) {
      err = "dst_address[" + std::to_string(k) + "]=" + std::to_string(dst1[k]) +
            " outside 1.." + std::to_string(n_dst);
      return false;
    }
  }

  // Two stable counting sorts (LSD): by src, then by dst -> order (dst, src, original k).
  std::vector<int32_t> by_src((size_t)nnz), order((size_t)nnz);
  {
    std::vector<int64_t> pos((size_t)n_src + 1, 0);
    for (int64_t k = 0; k < nnz; ++k) pos[(size_t)src1[k]]++;  // src1-1+1
    for (int64_t s = 0; s < n_src; ++s) pos[(size_t)s + 1] += pos[(size_t)s];
    for (int64_t k = 0; k < nnz; ++k) by_src[(size_t)pos[(size_t)src1[k] - 1]++] = (int32_t)k;
  }
  std::vector<int64_t> rawptr((size_t)n_dst + 1, 0);
  {
    for (int64_t k = 0; k < nnz; ++k) rawptr[(size_t)dst1[k]]++;
    for (int64_t d = 0; d < n_dst; ++d) rawptr[(size_t)d + 1] += rawptr[(size_t)d];
    std::vector<int64_t> pos(rawptr.begin(), rawptr.end() - 1);
    for (int64_t i = 0; i < nnz; ++i) {
      const int32_t k = by_src[(size_t)i];
      order[(size_t)pos[(size_t)dst1[k] - 1]++] = k;
    }
  }
  by_src.clear();
  by_src.shrink_to_fit();

  out.n_src = n_src;
  out.n_dst = n_dst;
  out.rowptr.assign((size_t)n_dst + 1, 0);
  out.col.clear();
  out.val.clear();
  out.col.reserve((size_t)nnz);
  out.val.reserve((size_t)nnz);
  out.max_row_nnz = 0;
  for (int64_t d = 0; d < n_dst; ++d) {
    const int64_t row_start = (int64_t)out.col.size();
    for (int64_t i = rawptr[(size_t)d]; i < rawptr[(size_t)d + 1]; ++i) {
      const int32_t k = order[(size_t)i];
      const int32_t s = src1[k] - 1;
      if ((int64_t)out.col.size() > row_start && out.col.back() == s) {
        out.val.back() += w[k];  // duplicate coordinate: summed in original link order
      } else {
        out.col.push_back(s);
        out.val.push_back(w[k]);
      }
    }
    out.rowptr[(size_t)d + 1] = (int64_t)out.col.size();
    out.max_row_nnz = std::max(out.max_row_nnz, (int64_t)out.col.size() - row_start);
  }
  out.nnz = (int64_t)out.col.size();

  std::vector<uint8_t> used((size_t)n_src, 0);
  for (int32_t c : out.col) used[(size_t)c] = 1;
  int64_t u = 0;
  for (uint8_t b : used) u += b;
  out.n_used_src = u;
  return true;
}

bool adopt_csr(int64_t n_src, int64_t n_dst, const int64_t* rowptr, const int32_t* col,
               const double* val, HostCsr& out, std::string& err) {
  if (n_src < 0 || n_dst < 0) {
    err = "negative size";
    return false;
  }
  if (n_src > INT32_MAX || n_dst > INT32_MAX) {
    err = "sizes beyond int32 addressing (SCRIP addresses are int32)";
    return false;
  }
  if (!rowptr) {
    err = "null rowptr";
    return false;
  }
  if (rowptr[0] != 0) {
    err = "rowptr[0] must be 0";
    return false;
  }
  for (int64_t d = 0; d < n_dst; ++d) {
    if (rowptr[d + 1] < rowptr[d]) {
      err = "rowptr decreases at row " + std::to_string(d);
      return false;
    }
  }
  const int64_t nnz = rowptr[n_dst];
  if (nnz > INT32_MAX) {
    err = "more than INT32_MAX links";
    return false;
  }
  if (nnz > 0 && (!col || !val)) {
    err = "null col/val array";
    return false;
  }
  std::vector<uint8_t> used((size_t)n_src, 0);
  int64_t max_row = 0;
  for (int64_t d = 0; d < n_dst; ++d) {
    int64_t prev = -1;
    for (int64_t i = rowptr[d]; i < rowptr[d + 1]; ++i) {
      const int64_t c = col[i];
      if (c < 0 || c >= n_src) {
        err = "col[" + std::to_string(i) + "]=" + std::to_string(c) + " outside 0.." + std::to_string(n_src - 1);
        return false;
      }
      if (c <= prev) {
        err = "row " + std::to_string(d) + ": columns must be strictly ascending (canonical CSR)";
        return false;
      }
      prev = c;
      used[(size_t)c] = 1;
    }
    max_row = std::max(max_row, rowptr[d + 1] - rowptr[d]);
  }
  out.n_src = n_src;
  out.n_dst = n_dst;
  out.nnz = nnz;
  out.max_row_nnz = max_row;
  out.rowptr.assign(rowptr, rowptr + n_dst + 1);
  out.col.assign(col, col + nnz);
  out.val.assign(val, val + nnz);
  int64_t u = 0;
  for (uint8_t b : used) u += b;
  out.n_used_src = u;
  return true;
}

int64_t prune_zero_links(HostCsr& csr) {
  int64_t kept = 0, max_row = 0;
  std::vector<int64_t> rowptr((size_t)csr.n_dst + 1, 0);
  for (int64_t d = 0; d < csr.n_dst; ++d) {
    const int64_t row_start = kept;
    for (int64_t i = csr.rowptr[(size_t)d]; i < csr.rowptr[(size_t)d + 1]; ++i) {
      if (csr.val[(size_t)i] == 0.0) continue;   // +0.0 and -0.0
      csr.col[(size_t)kept] = csr.col[(size_t)i];
      csr.val[(size_t)kept] = csr.val[(size_t)i];
      ++kept;
    }
    rowptr[(size_t)d + 1] = kept;
    max_row = std::max(max_row, kept - row_start);
  }
  const int64_t dropped = csr.nnz - kept;
  csr.rowptr.swap(rowptr);
  csr.col.resize((size_t)kept);
  csr.val.resize((size_t)kept);
  csr.nnz = kept;
  csr.max_row_nnz = max_row;
  std::vector<uint8_t> used((size_t)csr.n_src, 0);
  for (int32_t c : csr.col) used[(size_t)c] = 1;
  int64_t u = 0;
  for (uint8_t b : used) u += b;
  csr.n_used_src = u;
  return dropped;
}

HostChunk host_chunk_units(int64_t n_units, size_t x_unit, size_t y_unit, size_t xp_unit,
                           int64_t min_pack_units, int64_t pack_align, int64_t requested,
                           size_t free_bytes) {
  constexpr size_t kTarget = (size_t)256 << 20;   // ~256 MiB of X + Y per chunk
  constexpr size_t kPackCap = (size_t)1 << 30;    // the shortest packed chunk may not exceed 1 GiB
  HostChunk c;
  n_units = std::max<int64_t>(n_units, 1);
  min_pack_units = std::max<int64_t>(min_pack_units, 1);
  const size_t unit_w = std::max<size_t>(x_unit + y_unit, 1), unit_p = std::max<size_t>(xp_unit + y_unit, 1);
  c.pack = xp_unit > 0 && n_units >= min_pack_units;
  if (requested > 0) {   // the caller's chunk size is kept; too short a chunk does not pack
    if (requested < min_pack_units && requested < n_units) c.pack = false;
    c.units = std::min(requested, n_units);
    return c;
  }
  if (c.pack) {
    int64_t u = (int64_t)(kTarget / unit_p);
    if (u < min_pack_units) {
      // too few batch entries per chunk for the pack loops and the batch-fastest kernel: take the
      // minimum if that stays within the cap, else ship whole rows
      if ((size_t)min_pack_units * unit_p <= kPackCap) u = min_pack_units;
      else c.pack = false;
    }
    if (c.pack && pack_align > 1 && 2 * u >= pack_align)   // whole batch tiles of the kernel
      u = std::max<int64_t>(pack_align, u / pack_align * pack_align);
    c.units = u;
  }
  if (!c.pack) c.units = std::max<int64_t>(1, (int64_t)(kTarget / unit_w));
  c.units = std::min(c.units, n_units);
  // the four device buffers of a chunk (2 x X, 2 x Y) may take a quarter of the free memory at most
  auto clamp = [&](size_t unit) {
    if (free_bytes > 0 && 2 * (size_t)c.units * unit > ((size_t)2 << 30))
      c.units = std::max<int64_t>(1, std::min<int64_t>(c.units, (int64_t)(free_bytes / 4 / (unit + 1))));
  };
  clamp(c.pack ? unit_p : unit_w);
  if (c.pack && c.units < min_pack_units && c.units < n_units) {   // cannot pack in that little memory
    c.pack = false;
    c.units = std::min<int64_t>(n_units, std::max<int64_t>(1, (int64_t)(kTarget / unit_w)));
    clamp(unit_w);
  }
  return c;
}

void build_sell(const HostCsr& csr, HostSell& out) {
  const int64_t n_slices = (csr.n_dst + 63) / 64;
  out.n_slices = n_slices;
  out.slice_off.assign((size_t)n_slices + 1, 0);
  out.rowlen.assign((size_t)n_slices * 64, 0);
  for (int64_t d = 0; d < csr.n_dst; ++d)
    out.rowlen[(size_t)d] = (int32_t)(csr.rowptr[(size_t)d + 1] - csr.rowptr[(size_t)d]);
  for (int64_t s = 0; s < n_slices; ++s) {
    int32_t m = 0;
    for (int r = 0; r < 64; ++r) m = std::max(m, out.rowlen[(size_t)s * 64 + r]);
    out.slice_off[(size_t)s + 1] = out.slice_off[(size_t)s] + (int64_t)m * 64;
  }
  out.n_slots = out.slice_off[(size_t)n_slices];
  out.col.assign((size_t)out.n_slots, 0);
  out.val.assign((size_t)out.n_slots, 0.0);
  for (int64_t d = 0; d < csr.n_dst; ++d) {
    const int64_t s = d >> 6, r = d & 63;
    const int64_t base = out.slice_off[(size_t)s] + r;
    const int64_t p0 = csr.rowptr[(size_t)d];
    const int32_t len = out.rowlen[(size_t)d];
    for (int32_t k = 0; k < len; ++k) {
      out.col[(size_t)(base + (int64_t)k * 64)] = csr.col[(size_t)(p0 + k)];
      out.val[(size_t)(base + (int64_t)k * 64)] = csr.val[(size_t)(p0 + k)];
    }
    // padding slots of the slice repeat the row's last column (a cached, valid address)
    const int64_t nslots = (out.slice_off[(size_t)s + 1] - out.slice_off[(size_t)s]) / 64;
    const int32_t padcol = len > 0 ? csr.col[(size_t)(p0 + len - 1)] : 0;
    for (int64_t k = len; k < nslots; ++k) out.col[(size_t)(base + k * 64)] = padcol;
  }
}

void build_tile_plan(const HostCsr& csr, const HostSell& sell, int32_t rows_per_block_,
                     int32_t chunk_elems, int64_t max_chunks_per_block, HostTilePlan& plan) {
  plan = HostTilePlan();
  plan.rows_per_block = rows_per_block_;
  plan.chunk_elems = chunk_elems;
  if (csr.n_dst == 0 || rows_per_block_ <= 0 || chunk_elems <= 0) return;
  const int64_t rows_per_block = rows_per_block_;
  const int64_t n_blocks = (csr.n_dst + rows_per_block - 1) / rows_per_block;
  plan.n_blocks = n_blocks;
  plan.blk_chunk_off.assign((size_t)n_blocks + 1, 0);
  plan.blk_direct.assign((size_t)n_blocks, 0);
  plan.blk_lines.assign((size_t)n_blocks, 0);
  plan.lcol.assign((size_t)sell.n_slots, 0);
  constexpr int32_t kLineElems = 16;   // a 128-B line of f64

  std::vector<int32_t> chunks, cols;
  for (int64_t b = 0; b < n_blocks; ++b) {
    const int64_t d0 = b * rows_per_block;
    const int64_t d1 = std::min(csr.n_dst, d0 + rows_per_block);
    cols.assign(csr.col.begin() + csr.rowptr[(size_t)d0], csr.col.begin() + csr.rowptr[(size_t)d1]);
    std::sort(cols.begin(), cols.end());
    cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
    plan.total_distinct += (int64_t)cols.size();
    chunks.clear();
    for (int32_t c : cols) {
      const int32_t ch = c / chunk_elems;
      if (chunks.empty() || chunks.back() != ch) chunks.push_back(ch);
    }
    if ((int64_t)chunks.size() > max_chunks_per_block) {
      // too wide for LDS (polar caps of HEALPix targets, folds of tripolar grids ...): the kernel
      // gathers this block's links straight from X
      plan.blk_direct[(size_t)b] = 1;
      plan.direct_links += csr.rowptr[(size_t)d1] - csr.rowptr[(size_t)d0];
      plan.blk_chunk_off[(size_t)b + 1] = (int64_t)plan.chunk_src.size();
      continue;
    }
    plan.max_block_chunks = std::max(plan.max_block_chunks, (int64_t)chunks.size());
    {
      int32_t lines = 0, last = -1;
      for (int32_t c : cols) {
        if (c / kLineElems != last) {
          last = c / kLineElems;
          ++lines;
        }
      }
      plan.blk_lines[(size_t)b] = lines;
      plan.total_lines += lines;
    }
    const int64_t base = (int64_t)plan.chunk_src.size();
    plan.chunk_src.insert(plan.chunk_src.end(), chunks.begin(), chunks.end());
    plan.blk_chunk_off[(size_t)b + 1] = base + (int64_t)chunks.size();
    // LDS-local column of every link of the block
    for (int64_t d = d0; d < d1; ++d) {
      const int64_t s = d >> 6, r = d & 63;
      const int64_t sbase = sell.slice_off[(size_t)s] + r;
      const int64_t p0 = csr.rowptr[(size_t)d];
      const int32_t len = sell.rowlen[(size_t)d];
      for (int32_t k = 0; k < len; ++k) {
        const int32_t c = csr.col[(size_t)(p0 + k)];
        const int32_t ch = c / chunk_elems;
        const int64_t li =
            std::lower_bound(chunks.begin(), chunks.end(), ch) - chunks.begin();
        plan.lcol[(size_t)(sbase + (int64_t)k * 64)] =
            (int32_t)(li * chunk_elems + (c - ch * chunk_elems));
      }
    }
  }
  plan.total_chunks = (int64_t)plan.chunk_src.size();
  {
    std::vector<int32_t> all(plan.chunk_src);
    std::sort(all.begin(), all.end());
    plan.distinct_chunks = (int64_t)(std::unique(all.begin(), all.end()) - all.begin());
  }
  plan.valid = plan.direct_links * 4 <= csr.nnz;
}

#ifndef SMM_TIGHTEN_PCT
#define SMM_TIGHTEN_PCT 1   // per cent of the links that may sit in blocks demoted to direct gathering
#endif
int64_t tighten_tile_plan(const HostCsr& csr, HostTilePlan& plan, int64_t full_budget) {
  if (!plan.valid || plan.n_blocks == 0) return full_budget;
  const int64_t rows_per_block = plan.rows_per_block;
  auto block_links = [&](int64_t b) {
    const int64_t d0 = b * rows_per_block, d1 = std::min(csr.n_dst, d0 + rows_per_block);
    return csr.rowptr[(size_t)d1] - csr.rowptr[(size_t)d0];
  };
  int64_t chosen = full_budget;
  for (int64_t cand : {full_budget / 8, full_budget / 4, full_budget / 2}) {
    if (cand < 1 || cand >= plan.max_block_chunks) continue;   // would not shrink the tile
    int64_t demoted = 0;
    for (int64_t b = 0; b < plan.n_blocks; ++b)
      if (plan.blk_chunk_off[(size_t)b + 1] - plan.blk_chunk_off[(size_t)b] > cand) demoted += block_links(b);
    if ((plan.direct_links + demoted) * 100 <= csr.nnz * SMM_TIGHTEN_PCT) {
      chosen = cand;
      break;
    }
  }
  if (chosen == full_budget) return full_budget;
  std::vector<int32_t> kept;
  kept.reserve(plan.chunk_src.size());
  int64_t new_max = 0;
  int64_t prev_end = 0;  // old end offset of the previous block
  for (int64_t b = 0; b < plan.n_blocks; ++b) {
    const int64_t o0 = prev_end, o1 = plan.blk_chunk_off[(size_t)b + 1];
    prev_end = o1;
    plan.blk_chunk_off[(size_t)b] = (int64_t)kept.size();
    if (o1 - o0 > chosen) {
      plan.blk_direct[(size_t)b] = 1;
      plan.direct_links += block_links(b);
      plan.total_lines -= plan.blk_lines[(size_t)b];
      plan.blk_lines[(size_t)b] = 0;
    } else {
      kept.insert(kept.end(), plan.chunk_src.begin() + o0, plan.chunk_src.begin() + o1);
      new_max = std::max(new_max, o1 - o0);
    }
  }
  plan.blk_chunk_off[(size_t)plan.n_blocks] = (int64_t)kept.size();
  plan.chunk_src.swap(kept);
  plan.max_block_chunks = new_max;
  plan.total_chunks = (int64_t)plan.chunk_src.size();
  return chosen;
}

}  // namespace smm
