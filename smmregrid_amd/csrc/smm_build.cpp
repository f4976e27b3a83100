// Host-side operator construction (K6 of SURVEY.md section 2): SCRIP links ->
// canonical CSR -> SELL-64 device layout and the LDS source-tile plan.
//
// Replaces what the reference gets from the sparse.COO constructor
// (weights.py:31-42): 1-based addresses become 0-based, coordinates are sorted
// and duplicate coordinates are summed.
#include "smm_internal.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <thread>

namespace smm {

static std::atomic<int> g_host_threads{0};      // 0 = automatic
static std::atomic<int> g_active_builders{0};   // builders inside a parallel section right now
static std::atomic<bool> g_debug_no_threads{false};   // test hook: behave as if no thread could be started
static std::atomic<int64_t> g_debug_throw_in_task{-1};   // test hook: the n-th task body from now throws bad_alloc

void debug_builder_faults(bool no_threads, int64_t throw_in_task) {
  g_debug_no_threads.store(no_threads);
  g_debug_throw_in_task.store(throw_in_task);
}

int set_host_threads(int n) { return g_host_threads.exchange(n < 0 ? 0 : n); }
int fixed_host_threads() { return g_host_threads.load(); }

int host_threads(int64_t work_items, int64_t min_items_per_thread) {
  const int fixed = g_host_threads.load();
  if (fixed > 0) return fixed;   // the caller's count, whatever the work (tests force small inputs through every path)
  const int hw = std::min(16, usable_cpus());   // affinity capped by the cgroup quota, not the host's thread count
  const int share = std::max(1, hw / std::max(1, g_active_builders.load()));
  const int64_t by_work = std::max<int64_t>(1, work_items / std::max<int64_t>(1, min_items_per_thread));
  return (int)std::min<int64_t>(share, by_work);
}

namespace {

struct BuilderScope {   // counts this builder among the active ones for the automatic thread share
  BuilderScope() { g_active_builders.fetch_add(1); }
  ~BuilderScope() { g_active_builders.fetch_sub(1); }
};

// task(i) for i in [0, n): task 0 on the calling thread, the others on threads of their own.  Nothing a task
// throws (std::bad_alloc from a worker's scratch vectors) may leave its thread -- that would be std::terminate --
// so every body is wrapped, every started thread is joined, and the first exception is rethrown on the caller
// (create_operator maps it to a status).  A thread that cannot be started (std::system_error: EAGAIN under a
// pids / thread limit) costs parallelism, not the build: its task runs on the caller.
template <typename F>
void run_tasks(int n, F task) {
  if (n <= 0) return;
  std::exception_ptr first;
  std::mutex mu;
  auto guarded = [&](int i) {
    try {
      if (g_debug_throw_in_task.load(std::memory_order_relaxed) >= 0 && g_debug_throw_in_task.fetch_sub(1) == 0)
        throw std::bad_alloc();
      task(i);
    } catch (...) {
      std::lock_guard<std::mutex> lock(mu);
      if (!first) first = std::current_exception();
    }
  };
  std::vector<std::thread> pool;
  std::vector<int> on_caller;
  try {
    pool.reserve((size_t)n - 1);
  } catch (...) {   // not even the handles: everything runs here
  }
  for (int i = 1; i < n; ++i) {
    bool started = false;
    if (pool.size() < pool.capacity() && !g_debug_no_threads.load(std::memory_order_relaxed)) {
      try {
        pool.emplace_back(guarded, i);
        started = true;
      } catch (...) {
      }
    }
    if (!started) {
      try {
        on_caller.push_back(i);
      } catch (...) {
        guarded(i);   // run it right away rather than lose it
      }
    }
  }
  guarded(0);
  for (int i : on_caller) guarded(i);
  for (auto& th : pool) th.join();
  if (first) std::rethrow_exception(first);
}

// fn(t, lo, hi) over [0, n) cut into `nt` contiguous ranges; the calling thread takes range 0.
template <typename F>
void parallel_ranges(int64_t n, int nt, F fn) {
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, n));
  if (nt == 1) {
    fn(0, (int64_t)0, n);
    return;
  }
  run_tasks(nt, [&](int t) { fn(t, n * t / nt, n * (t + 1) / nt); });
}

}  // namespace

namespace {

inline void mark_byte(uint8_t* p) { __atomic_store_n(p, (uint8_t)1, __ATOMIC_RELAXED); }

// n_used_src and max_row_nnz of a finished CSR (parallel over links / rows).
void finish_csr_stats(HostCsr& out, int nt) {
  std::vector<uint8_t> used((size_t)out.n_src, 0);
  parallel_ranges(out.nnz, nt, [&](int, int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) mark_byte(&used[(size_t)out.col[(size_t)i]]);
  });
  std::vector<int64_t> part_u((size_t)nt, 0), part_m((size_t)nt, 0);
  parallel_ranges(out.n_src, nt, [&](int t, int64_t lo, int64_t hi) {
    int64_t u = 0;
    for (int64_t s = lo; s < hi; ++s) u += used[(size_t)s];
    part_u[(size_t)t] = u;
  });
  parallel_ranges(out.n_dst, nt, [&](int t, int64_t lo, int64_t hi) {
    int64_t m = 0;
    for (int64_t d = lo; d < hi; ++d) m = std::max(m, out.rowptr[(size_t)d + 1] - out.rowptr[(size_t)d]);
    part_m[(size_t)t] = m;
  });
  out.n_used_src = 0;
  out.max_row_nnz = 0;
  for (int t = 0; t < nt; ++t) {
    out.n_used_src += part_u[(size_t)t];
    out.max_row_nnz = std::max(out.max_row_nnz, part_m[(size_t)t]);
  }
}

}  // namespace

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
  BuilderScope scope;
  const int nt = host_threads(nnz, (int64_t)1 << 17);

  // pass 1: addresses in range (the lowest bad index is reported, src before dst) and, on the way,
  // whether the links already come ordered by (dst, src) -- what `cdo gen*` writes
  std::vector<int64_t> first_bad((size_t)nt, nnz);
  std::vector<uint8_t> unsorted((size_t)nt, 0);
  parallel_ranges(nnz, nt, [&](int t, int64_t lo, int64_t hi) {
    uint8_t uns = 0;
    for (int64_t k = lo; k < hi; ++k) {
      const int64_t s = (int64_t)src1[k] - 1, d = (int64_t)dst1[k] - 1;
      if (s < 0 || s >= n_src || d < 0 || d >= n_dst) {
        first_bad[(size_t)t] = k;
        break;
      }
      if (k > 0 && (dst1[k - 1] > dst1[k] || (dst1[k - 1] == dst1[k] && src1[k - 1] > src1[k]))) uns = 1;
    }
    unsorted[(size_t)t] = uns;
  });
  {
    const int64_t k = *std::min_element(first_bad.begin(), first_bad.end());
    if (k < nnz) {
      const int64_t s = (int64_t)src1[k] - 1;
      if (s < 0 || s >= n_src)
        err = "src_address[" + std::to_string(k) + "]=" + std::to_string(src1[k]) + " outside 1.." + std::to_string(n_src);
      else
        err = "dst_address[" + std::to_string(k) + "]=" + std::to_string(dst1[k]) + " outside 1.." + std::to_string(n_dst);
      return false;
    }
  }
  const bool sorted = std::none_of(unsorted.begin(), unsorted.end(), [](uint8_t u) { return u != 0; });

  out.n_src = n_src;
  out.n_dst = n_dst;
  out.rowptr.assign((size_t)n_dst + 1, 0);
  out.col.clear();
  out.val.clear();

  auto same = [&](int64_t a, int64_t b) { return dst1[a] == dst1[b] && src1[a] == src1[b]; };

  if (sorted) {
    // Ordered input: a coordinate is a run of equal (dst, src); a run is summed in link order by the
    // thread its first link belongs to (range starts are moved to run starts).
    std::vector<int64_t> lo_of((size_t)nt + 1, nnz), kept((size_t)nt + 1, 0);
    for (int t = 0; t < nt; ++t) {
      int64_t lo = nnz * t / nt;
      while (lo > 0 && lo < nnz && same(lo, lo - 1)) ++lo;
      lo_of[(size_t)t] = lo;
    }
    for (int t = nt - 1; t > 0; --t) lo_of[(size_t)t] = std::max(lo_of[(size_t)t], lo_of[(size_t)t - 1]);
    lo_of[0] = 0;
    auto count = [&](int t) {
      int64_t n = 0;
      for (int64_t k = lo_of[(size_t)t]; k < lo_of[(size_t)t + 1]; ++k) n += (k == lo_of[(size_t)t] || !same(k, k - 1));
      kept[(size_t)t + 1] = n;
    };
    run_tasks(nt, count);
    for (int t = 0; t < nt; ++t) kept[(size_t)t + 1] += kept[(size_t)t];
    const int64_t total = kept[(size_t)nt];
    out.col.resize((size_t)total);
    out.val.resize((size_t)total);
    auto fill = [&](int t) {
      const int64_t lo = lo_of[(size_t)t], hi = lo_of[(size_t)t + 1];
      int64_t p = kept[(size_t)t] - 1;
      // rowptr[r] = kept entries of rows < r: written where the row index steps up
      int64_t prev_row = lo > 0 ? (int64_t)dst1[lo - 1] - 1 : -1;
      for (int64_t k = lo; k < hi; ++k) {
        if (k == lo || !same(k, k - 1)) {
          ++p;
          out.col[(size_t)p] = src1[k] - 1;
          out.val[(size_t)p] = w[k];
          const int64_t row = (int64_t)dst1[k] - 1;
          for (int64_t r = prev_row + 1; r <= row; ++r) out.rowptr[(size_t)r] = p;
          prev_row = row;
        } else {
          out.val[(size_t)p] += w[k];   // duplicate coordinate: summed in original link order
        }
      }
    };
    run_tasks(nt, fill);
    for (int64_t r = (nnz > 0 ? (int64_t)dst1[nnz - 1] : 0); r <= n_dst; ++r) out.rowptr[(size_t)r] = total;  // rows after the last link
    out.nnz = total;
    finish_csr_stats(out, nt);
    return true;
  }

  // General input.  Links are dealt into `nb` buckets of consecutive destination rows (stable: thread
  // ranges in order, links of a range in order); every bucket is then ordered by (dst, src, link index)
  // by its own thread -- a stable counting sort by row, a stable sort by source cell inside each row --
  // and its duplicate coordinates are summed in that order.
  const int nb = nt;
  auto bucket_of = [&](int32_t d1) { return (int)(((int64_t)d1 - 1) * nb / std::max<int64_t>(n_dst, 1)); };
  std::vector<int64_t> cnt((size_t)nt * nb, 0);
  parallel_ranges(nnz, nt, [&](int t, int64_t lo, int64_t hi) {
    int64_t* c = &cnt[(size_t)t * nb];
    for (int64_t k = lo; k < hi; ++k) ++c[bucket_of(dst1[k])];
  });
  std::vector<int64_t> bstart((size_t)nb + 1, 0), off((size_t)nt * nb, 0);
  for (int b = 0; b < nb; ++b) {
    int64_t run = bstart[(size_t)b];
    for (int t = 0; t < nt; ++t) {
      off[(size_t)t * nb + b] = run;
      run += cnt[(size_t)t * nb + b];
    }
    bstart[(size_t)b + 1] = run;
  }
  std::vector<int32_t> idx((size_t)nnz), order((size_t)nnz);
  parallel_ranges(nnz, nt, [&](int t, int64_t lo, int64_t hi) {
    int64_t* o = &off[(size_t)t * nb];
    for (int64_t k = lo; k < hi; ++k) idx[(size_t)o[bucket_of(dst1[k])]++] = (int32_t)k;
  });
  std::vector<std::vector<int32_t>> loc_col((size_t)nb);
  std::vector<std::vector<double>> loc_val((size_t)nb);
  {
    auto work = [&](int b) {
      // rows of this bucket: floor(d * nb / n_dst) == b  <=>  ceil(b * n_dst / nb) <= d < ceil((b + 1) * n_dst / nb)
      const int64_t r0 = (n_dst * b + nb - 1) / nb, r1 = (n_dst * (b + 1) + nb - 1) / nb;
      const int64_t i0 = bstart[(size_t)b], i1 = bstart[(size_t)b + 1];
      std::vector<int64_t> pos((size_t)(r1 - r0) + 1, 0);
      for (int64_t i = i0; i < i1; ++i) ++pos[(size_t)(dst1[idx[(size_t)i]] - 1 - r0) + 1];
      for (int64_t r = 0; r < r1 - r0; ++r) pos[(size_t)r + 1] += pos[(size_t)r];
      std::vector<int64_t> row_end(pos.begin() + 1, pos.end());
      {
        std::vector<int64_t> cur(pos.begin(), pos.end() - 1);
        for (int64_t i = i0; i < i1; ++i) {
          const int32_t k = idx[(size_t)i];
          order[(size_t)(i0 + cur[(size_t)(dst1[k] - 1 - r0)]++)] = k;
        }
      }
      std::vector<int32_t>& lc = loc_col[(size_t)b];
      std::vector<double>& lv = loc_val[(size_t)b];
      lc.reserve((size_t)(i1 - i0));
      lv.reserve((size_t)(i1 - i0));
      for (int64_t r = 0; r < r1 - r0; ++r) {
        int32_t* a = &order[(size_t)(i0 + (r ? row_end[(size_t)r - 1] : 0))];
        const int64_t len = row_end[(size_t)r] - (r ? row_end[(size_t)r - 1] : 0);
        if (len <= 32) {   // stable insertion sort by source cell
          for (int64_t i = 1; i < len; ++i) {
            const int32_t k = a[i];
            const int32_t s = src1[k];
            int64_t j = i;
            while (j > 0 && src1[a[j - 1]] > s) {
              a[j] = a[j - 1];
              --j;
            }
            a[j] = k;
          }
        } else {
          std::stable_sort(a, a + len, [&](int32_t x, int32_t y) { return src1[x] < src1[y]; });
        }
        const size_t row_start = lc.size();
        for (int64_t i = 0; i < len; ++i) {
          const int32_t k = a[i];
          const int32_t s = src1[k] - 1;
          if (lc.size() > row_start && lc.back() == s) {
            lv.back() += w[k];  // duplicate coordinate: summed in original link order
          } else {
            lc.push_back(s);
            lv.push_back(w[k]);
          }
        }
        out.rowptr[(size_t)(r0 + r) + 1] = (int64_t)(lc.size() - row_start);   // row length for now
      }
    };
    run_tasks(nb, work);
  }
  idx.clear();
  idx.shrink_to_fit();
  order.clear();
  order.shrink_to_fit();
  for (int64_t d = 0; d < n_dst; ++d) out.rowptr[(size_t)d + 1] += out.rowptr[(size_t)d];
  out.nnz = out.rowptr[(size_t)n_dst];
  out.col.resize((size_t)out.nnz);
  out.val.resize((size_t)out.nnz);
  {
    std::vector<int64_t> at((size_t)nb + 1, 0);
    for (int b = 0; b < nb; ++b) at[(size_t)b + 1] = at[(size_t)b] + (int64_t)loc_col[(size_t)b].size();
    auto copy = [&](int b) {
      if (loc_col[(size_t)b].empty()) return;
      memcpy(&out.col[(size_t)at[(size_t)b]], loc_col[(size_t)b].data(), loc_col[(size_t)b].size() * sizeof(int32_t));
      memcpy(&out.val[(size_t)at[(size_t)b]], loc_val[(size_t)b].data(), loc_val[(size_t)b].size() * sizeof(double));
    };
    run_tasks(nb, copy);
  }
  finish_csr_stats(out, nt);
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
  constexpr size_t kMinChunk = (size_t)32 << 20;  // ... and no chunk is cut below 32 MiB to reach kMinChunks per call
  constexpr int64_t kMinChunks = 8;
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
    if (c.pack) {
      // The first chunk's pack and the last chunk's D2H + copy-out overlap nothing: a call is cut into at least
      // kMinChunks chunks while a chunk keeps kMinChunk bytes and the pack's minimum (config 2, 512 rows from host
      // memory: 4 chunks of 128 rows 29.7 ms, 8 of 64 rows -- see profiles/r06_host_to_host.txt)
      const int64_t floor_units = std::max<int64_t>(min_pack_units, (int64_t)(kMinChunk / unit_p));
      const int64_t cap = std::max(floor_units, (n_units + kMinChunks - 1) / kMinChunks);
      if (u > cap) u = cap >= 32 ? cap / 32 * 32 : cap;
    }
    c.units = u;
  }
  if (!c.pack) {
    c.units = std::max<int64_t>(1, (int64_t)(kTarget / unit_w));
    const int64_t floor_units = std::max<int64_t>(1, (int64_t)(kMinChunk / unit_w));
    c.units = std::min(c.units, std::max(floor_units, (n_units + kMinChunks - 1) / kMinChunks));
  }
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
  BuilderScope scope;
  const int nt = host_threads(csr.nnz + csr.n_dst, (int64_t)1 << 17);
  const int64_t n_slices = (csr.n_dst + 63) / 64;
  out.n_slices = n_slices;
  out.slice_off.assign((size_t)n_slices + 1, 0);
  out.rowlen.assign((size_t)n_slices * 64, 0);
  parallel_ranges(n_slices, nt, [&](int, int64_t lo, int64_t hi) {
    for (int64_t s = lo; s < hi; ++s) {
      int32_t m = 0;
      for (int64_t d = s * 64; d < std::min(csr.n_dst, s * 64 + 64); ++d) {
        const int32_t len = (int32_t)(csr.rowptr[(size_t)d + 1] - csr.rowptr[(size_t)d]);
        out.rowlen[(size_t)d] = len;
        m = std::max(m, len);
      }
      out.slice_off[(size_t)s + 1] = (int64_t)m * 64;   // slots of the slice for now
    }
  });
  for (int64_t s = 0; s < n_slices; ++s) out.slice_off[(size_t)s + 1] += out.slice_off[(size_t)s];
  out.n_slots = out.slice_off[(size_t)n_slices];
  out.col.resize((size_t)out.n_slots);
  out.val.resize((size_t)out.n_slots);
  parallel_ranges(n_slices, nt, [&](int, int64_t lo, int64_t hi) {
    for (int64_t s = lo; s < hi; ++s) {
      const int64_t nslots = (out.slice_off[(size_t)s + 1] - out.slice_off[(size_t)s]) / 64;
      for (int64_t r = 0; r < 64; ++r) {
        const int64_t d = s * 64 + r;
        const int64_t base = out.slice_off[(size_t)s] + r;
        const int32_t len = out.rowlen[(size_t)d];
        const int64_t p0 = d < csr.n_dst ? csr.rowptr[(size_t)d] : 0;
        for (int32_t k = 0; k < len; ++k) {
          out.col[(size_t)(base + (int64_t)k * 64)] = csr.col[(size_t)(p0 + k)];
          out.val[(size_t)(base + (int64_t)k * 64)] = csr.val[(size_t)(p0 + k)];
        }
        // padding slots of the slice repeat the row's last column (a cached, valid address), weight 0
        const int32_t padcol = len > 0 ? csr.col[(size_t)(p0 + len - 1)] : 0;
        for (int64_t k = len; k < nslots; ++k) {
          out.col[(size_t)(base + k * 64)] = padcol;
          out.val[(size_t)(base + k * 64)] = 0.0;
        }
      }
    }
  });
}

void build_tile_plan(const HostCsr& csr, const HostSell& sell, int32_t rows_per_block_,
                     int32_t chunk_elems, int64_t max_chunks_per_block, HostTilePlan& plan) {
  plan = HostTilePlan();
  plan.rows_per_block = rows_per_block_;
  plan.chunk_elems = chunk_elems;
  if (csr.n_dst == 0 || rows_per_block_ <= 0 || chunk_elems <= 0) return;
  BuilderScope scope;
  const int64_t rows_per_block = rows_per_block_;
  const int64_t n_blocks = (csr.n_dst + rows_per_block - 1) / rows_per_block;
  const int nt = host_threads(csr.nnz + csr.n_dst, (int64_t)1 << 16);
  plan.n_blocks = n_blocks;
  plan.blk_chunk_off.assign((size_t)n_blocks + 1, 0);
  plan.blk_direct.assign((size_t)n_blocks, 0);
  plan.blk_lines.assign((size_t)n_blocks, 0);
  plan.lcol.resize((size_t)sell.n_slots);
  constexpr int32_t kLineElems = 16;   // a 128-B line of f64

  // Blocks are independent: thread t plans a contiguous range of them into its own chunk list (the lists
  // are concatenated in block order afterwards) and writes the LDS-local columns of its blocks' links.
  struct Part {
    std::vector<int32_t> chunk_src;
    int64_t total_distinct = 0, direct_links = 0, max_block_chunks = 0, total_lines = 0;
  };
  std::vector<Part> parts((size_t)nt);
  std::vector<uint8_t> chunk_seen((size_t)(csr.n_src / chunk_elems) + 1, 0);   // distinct chunks over the operator
  parallel_ranges(n_blocks, nt, [&](int t, int64_t b_lo, int64_t b_hi) {
    Part& me = parts[(size_t)t];
    std::vector<int32_t> chunks, cols;
    for (int64_t b = b_lo; b < b_hi; ++b) {
      const int64_t d0 = b * rows_per_block;
      const int64_t d1 = std::min(csr.n_dst, d0 + rows_per_block);
      cols.assign(csr.col.begin() + csr.rowptr[(size_t)d0], csr.col.begin() + csr.rowptr[(size_t)d1]);
      std::sort(cols.begin(), cols.end());
      cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
      me.total_distinct += (int64_t)cols.size();
      chunks.clear();
      for (int32_t c : cols) {
        const int32_t ch = c / chunk_elems;
        if (chunks.empty() || chunks.back() != ch) chunks.push_back(ch);
      }
      // lcol slots of rows past a row's length (SELL padding) stay 0 as before
      for (int64_t d = d0; d < d1; ++d) {
        const int64_t s = d >> 6, r = d & 63;
        const int64_t sbase = sell.slice_off[(size_t)s] + r;
        const int64_t nslots = (sell.slice_off[(size_t)s + 1] - sell.slice_off[(size_t)s]) / 64;
        for (int64_t k = 0; k < nslots; ++k) plan.lcol[(size_t)(sbase + k * 64)] = 0;
      }
      if ((int64_t)chunks.size() > max_chunks_per_block) {
        // too wide for LDS (polar caps of HEALPix targets, folds of tripolar grids ...): the kernel
        // gathers this block's links straight from X
        plan.blk_direct[(size_t)b] = 1;
        me.direct_links += csr.rowptr[(size_t)d1] - csr.rowptr[(size_t)d0];
        continue;   // blk_chunk_off[b + 1] holds the block's chunk count (0) until the prefix sum
      }
      me.max_block_chunks = std::max(me.max_block_chunks, (int64_t)chunks.size());
      {
        int32_t lines = 0, last = -1;
        for (int32_t c : cols) {
          if (c / kLineElems != last) {
            last = c / kLineElems;
            ++lines;
          }
        }
        plan.blk_lines[(size_t)b] = lines;
        me.total_lines += lines;
      }
      me.chunk_src.insert(me.chunk_src.end(), chunks.begin(), chunks.end());
      plan.blk_chunk_off[(size_t)b + 1] = (int64_t)chunks.size();
      for (int32_t ch : chunks) mark_byte(&chunk_seen[(size_t)ch]);
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
  });
  // rows of the last slice past n_dst belong to no block: their slots stay 0 too
  for (int64_t d = csr.n_dst; d < sell.n_slices * 64; ++d) {
    const int64_t s = d >> 6, r = d & 63;
    const int64_t nslots = (sell.slice_off[(size_t)s + 1] - sell.slice_off[(size_t)s]) / 64;
    for (int64_t k = 0; k < nslots; ++k) plan.lcol[(size_t)(sell.slice_off[(size_t)s] + r + k * 64)] = 0;
  }
  for (int64_t b = 0; b < n_blocks; ++b) plan.blk_chunk_off[(size_t)b + 1] += plan.blk_chunk_off[(size_t)b];
  plan.total_chunks = plan.blk_chunk_off[(size_t)n_blocks];
  plan.chunk_src.resize((size_t)plan.total_chunks);
  {
    std::vector<int64_t> at((size_t)nt + 1, 0);
    for (int t = 0; t < nt; ++t) {
      at[(size_t)t + 1] = at[(size_t)t] + (int64_t)parts[(size_t)t].chunk_src.size();
      plan.total_distinct += parts[(size_t)t].total_distinct;
      plan.direct_links += parts[(size_t)t].direct_links;
      plan.total_lines += parts[(size_t)t].total_lines;
      plan.max_block_chunks = std::max(plan.max_block_chunks, parts[(size_t)t].max_block_chunks);
    }
    parallel_ranges(nt, nt, [&](int, int64_t lo, int64_t hi) {
      for (int64_t t = lo; t < hi; ++t)
        if (!parts[(size_t)t].chunk_src.empty())
          memcpy(&plan.chunk_src[(size_t)at[(size_t)t]], parts[(size_t)t].chunk_src.data(),
                 parts[(size_t)t].chunk_src.size() * sizeof(int32_t));
    });
  }
  {
    std::vector<int64_t> part((size_t)nt, 0);
    parallel_ranges((int64_t)chunk_seen.size(), nt, [&](int t, int64_t lo, int64_t hi) {
      int64_t n = 0;
      for (int64_t i = lo; i < hi; ++i) n += chunk_seen[(size_t)i];
      part[(size_t)t] = n;
    });
    for (int64_t n : part) plan.distinct_chunks += n;
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
