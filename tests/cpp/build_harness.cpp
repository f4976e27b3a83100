// Host-only harness around smm_build.cpp (COO -> CSR -> SELL-64 -> LDS tile plans), compiled
// with g++ -fsanitize=address,undefined by tests/test_build_sanitized.py.
// argv[1]: host threads of the builders (0 / absent = automatic)
// stdin : n_src n_dst nnz, then nnz lines "src1 dst1 w"
// stdout: the canonical CSR and invariants of the SELL / tile-plan structures.
#include <cinttypes>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../smmregrid_amd/csrc/smm_internal.h"

int main(int argc, char** argv) {
  if (argc > 1) smm::set_host_threads(atoi(argv[1]));
  long long n_src, n_dst, nnz;
  if (scanf("%lld %lld %lld", &n_src, &n_dst, &nnz) != 3) return 2;
  std::vector<int32_t> src((size_t)nnz), dst((size_t)nnz);
  std::vector<double> w((size_t)nnz);
  for (long long k = 0; k < nnz; ++k)
    if (scanf("%d %d %lf", &src[(size_t)k], &dst[(size_t)k], &w[(size_t)k]) != 3) return 2;
  smm::HostCsr csr;
  std::string err;
  if (!smm::build_csr(n_src, n_dst, nnz, src.data(), dst.data(), w.data(), csr, err)) {
    printf("ERROR %s\n", err.c_str());
    return 0;
  }
  smm::HostSell sell;
  smm::build_sell(csr, sell);
  printf("CSR %lld %lld %lld %lld\n", (long long)csr.nnz, (long long)csr.n_used_src,
         (long long)csr.max_row_nnz, (long long)sell.n_slots);
  for (int64_t d = 0; d <= csr.n_dst; ++d) printf("%lld ", (long long)csr.rowptr[(size_t)d]);
  printf("\n");
  for (int64_t p = 0; p < csr.nnz; ++p) printf("%d ", csr.col[(size_t)p]);
  printf("\n");
  for (int64_t p = 0; p < csr.nnz; ++p) printf("%.17g ", csr.val[(size_t)p]);
  printf("\n");
  // SELL invariants: every link sits at slice_off + k*64 + lane, padding repeats the last column
  long long bad = 0;
  for (int64_t d = 0; d < csr.n_dst; ++d) {
    const int64_t s = d >> 6, r = d & 63;
    const int64_t base = sell.slice_off[(size_t)s] + r;
    const int64_t nslots = (sell.slice_off[(size_t)s + 1] - sell.slice_off[(size_t)s]) / 64;
    const int32_t len = sell.rowlen[(size_t)d];
    if (len != csr.rowptr[(size_t)d + 1] - csr.rowptr[(size_t)d] || len > nslots) ++bad;
    for (int64_t k = 0; k < nslots; ++k) {
      const int32_t c = sell.col[(size_t)(base + k * 64)];
      const double v = sell.val[(size_t)(base + k * 64)];
      if (k < len) {
        if (c != csr.col[(size_t)(csr.rowptr[(size_t)d] + k)] || v != csr.val[(size_t)(csr.rowptr[(size_t)d] + k)]) ++bad;
      } else {
        if (v != 0.0 || c < 0 || (n_src > 0 && c >= n_src)) ++bad;
      }
    }
  }
  // both tile-plan shapes: every link's LDS index must resolve to its source column
  for (int rows : {256, 64, 16}) {
    const int spb = rows / 64;   // 0 for the sub-slice shape
    smm::HostTilePlan plan;
    const int64_t budget = rows == 256 ? 512 : 128;
    smm::build_tile_plan(csr, sell, rows, 16, budget, plan);
    auto check_plan = [&](int64_t limit) {
      long long pbad = 0;
      if (!plan.valid) return pbad;
      const int64_t rows_per_block = rows;
      int64_t direct_links = 0;
      for (int64_t b = 0; b < plan.n_blocks; ++b) {
        const int64_t nch = plan.blk_chunk_off[(size_t)b + 1] - plan.blk_chunk_off[(size_t)b];
        if (nch < 0 || nch > limit || nch > plan.max_block_chunks) ++pbad;
        if (plan.blk_direct[(size_t)b]) {
          if (nch != 0) ++pbad;
          const int64_t d0 = b * rows_per_block, d1 = std::min(csr.n_dst, d0 + rows_per_block);
          direct_links += csr.rowptr[(size_t)d1] - csr.rowptr[(size_t)d0];
        }
      }
      if (direct_links != plan.direct_links || plan.total_chunks != (int64_t)plan.chunk_src.size()) ++pbad;
      for (int64_t d = 0; d < csr.n_dst; ++d) {
        const int64_t b = d / rows_per_block, s = d >> 6, r = d & 63;
        if (plan.blk_direct[(size_t)b]) continue;  // gathered from X directly, no LDS indices
        const int64_t c0 = plan.blk_chunk_off[(size_t)b];
        const int64_t base = sell.slice_off[(size_t)s] + r;
        for (int32_t k = 0; k < sell.rowlen[(size_t)d]; ++k) {
          const int32_t li = plan.lcol[(size_t)(base + (int64_t)k * 64)];
          const int64_t ch = li / 16, e = li % 16;
          if (c0 + ch >= plan.blk_chunk_off[(size_t)b + 1]) { ++pbad; continue; }
          const int64_t col = (int64_t)plan.chunk_src[(size_t)(c0 + ch)] * 16 + e;
          if (col != csr.col[(size_t)(csr.rowptr[(size_t)d] + k)]) ++pbad;
        }
      }
      return pbad;
    };
    long long pbad = check_plan(budget);
    printf("PLAN %d %d %lld %lld %lld %lld %lld\n", spb, (int)plan.valid, (long long)plan.max_block_chunks,
           (long long)plan.total_chunks, (long long)plan.distinct_chunks, (long long)plan.direct_links, pbad);
    // tightened budget: demoted blocks are direct, every remaining LDS index still resolves
    const int64_t chosen = smm::tighten_tile_plan(csr, plan, budget);
    long long tbad = check_plan(chosen);
    if (plan.valid && plan.direct_links * 100 > csr.nnz && chosen != budget) ++tbad;
    printf("TIGHT %d %lld %lld %lld %lld\n", spb, (long long)chosen, (long long)plan.max_block_chunks,
           (long long)plan.direct_links, tbad);
  }
  // adopt_csr: the canonical CSR round-trips unchanged; each corruption of it is rejected
  {
    long long abad = 0;
    smm::HostCsr back;
    std::string e2;
    if (!smm::adopt_csr(n_src, n_dst, csr.rowptr.data(), csr.col.data(), csr.val.data(), back, e2)) ++abad;
    if (back.rowptr != csr.rowptr || back.col != csr.col || back.val != csr.val || back.nnz != csr.nnz ||
        back.n_used_src != csr.n_used_src || back.max_row_nnz != csr.max_row_nnz)
      ++abad;
    smm::HostCsr tmp;
    if (csr.nnz > 0) {
      std::vector<int32_t> c2(csr.col);
      c2[0] = (int32_t)n_src;                                   // column out of range
      if (smm::adopt_csr(n_src, n_dst, csr.rowptr.data(), c2.data(), csr.val.data(), tmp, e2)) ++abad;
      std::vector<int64_t> r2(csr.rowptr);
      r2[0] = 1;                                                // rowptr must start at 0
      if (smm::adopt_csr(n_src, n_dst, r2.data(), csr.col.data(), csr.val.data(), tmp, e2)) ++abad;
    }
    for (int64_t d = 0; d < csr.n_dst; ++d) {
      if (csr.rowptr[(size_t)d + 1] - csr.rowptr[(size_t)d] >= 2) {
        std::vector<int32_t> c2(csr.col);
        std::swap(c2[(size_t)csr.rowptr[(size_t)d]], c2[(size_t)csr.rowptr[(size_t)d] + 1]);  // unsorted row
        if (smm::adopt_csr(n_src, n_dst, csr.rowptr.data(), c2.data(), csr.val.data(), tmp, e2)) ++abad;
        c2 = csr.col;
        c2[(size_t)csr.rowptr[(size_t)d] + 1] = c2[(size_t)csr.rowptr[(size_t)d]];            // repeated column
        if (smm::adopt_csr(n_src, n_dst, csr.rowptr.data(), c2.data(), csr.val.data(), tmp, e2)) ++abad;
        std::vector<int64_t> r2(csr.rowptr);
        r2[(size_t)d + 1] = r2[(size_t)d] - 1;                                                  // decreasing rowptr
        if (smm::adopt_csr(n_src, n_dst, r2.data(), csr.col.data(), csr.val.data(), tmp, e2)) ++abad;
        break;
      }
    }
    printf("ADOPTBAD %lld\n", abad);
  }
  // chunk sizing of the host pipelines (smm_apply_host / smm_group_apply_host): Y counts too
  {
    long long cbad = 0;
    const size_t GiB = (size_t)1 << 30, MiB = (size_t)1 << 20;
    // config 2: 3600 rows, S = 1 038 240 f64 rows on 128-B lines, D = 64 800, U = 259 200
    smm::HostChunk c = smm::host_chunk_units(3600, 8305920, 518400, 2073600, 32, 128, 0, 200 * GiB);
    if (!c.pack || c.units % 128 != 0 || c.units < 128 || (size_t)c.units * (2073600 + 518400) > 512 * MiB) ++cbad;
    // U << D: 64 used source cells feeding a 12.6-M-cell target (a regional subset onto HEALPix): the
    // chunk is bound by its Y bytes, not by the packed X
    c = smm::host_chunk_units(4096, 40000 * 8, (size_t)12582912 * 8, 64 * 8, 32, 128, 0, 200 * GiB);
    if (c.units < 1 || (size_t)c.units * ((size_t)12582912 * 8 + 64 * 8) > GiB) ++cbad;
    if (c.pack && c.units < 32) ++cbad;
    // the same with little free memory: four buffers stay within a quarter of it
    c = smm::host_chunk_units(4096, 40000 * 8, (size_t)12582912 * 8, 64 * 8, 32, 128, 0, 4 * GiB);
    if (c.units < 1 || 4 * (size_t)c.units * ((c.pack ? 64 * 8 : 40000 * 8) + (size_t)12582912 * 8) > 4 * GiB + 4 * ((size_t)12582912 * 8)) ++cbad;
    if (c.pack && c.units < 32) ++cbad;
    // short batches never pack; a requested chunk size is kept
    c = smm::host_chunk_units(16, 8000, 800, 2000, 32, 128, 0, 0);
    if (c.pack || c.units != 16) ++cbad;
    c = smm::host_chunk_units(1000, 8000, 800, 2000, 32, 128, 100, 0);
    if (!c.pack || c.units != 100) ++cbad;
    c = smm::host_chunk_units(1000, 8000, 800, 2000, 32, 128, 7, 0);
    if (c.pack || c.units != 7) ++cbad;
    c = smm::host_chunk_units(1000, 8000, 800, 0, 32, 128, 0, 0);
    if (c.pack || c.units != 1000) ++cbad;
    // a level group whose 32 batch entries of all levels exceed 1 GiB of staging keeps whole rows
    c = smm::host_chunk_units(120, (size_t)75 * 11778304, (size_t)75 * 518400, (size_t)39000000 * 8, 32, 1, 0, 200 * GiB);
    if (c.pack || c.units < 1) ++cbad;
    printf("CHUNKBAD %lld\n", cbad);
  }
  // exact-zero links dropped: what is left is the same matrix without its zeros
  {
    long long pbad = 0;
    smm::HostCsr pr = csr;
    const int64_t dropped = smm::prune_zero_links(pr);
    int64_t zeros = 0;
    for (double v : csr.val) zeros += v == 0.0;
    if (dropped != zeros || pr.nnz != csr.nnz - zeros || (int64_t)pr.col.size() != pr.nnz || pr.rowptr.back() != pr.nnz) ++pbad;
    int64_t max_row = 0;
    for (int64_t d = 0; d < csr.n_dst; ++d) {
      int64_t k2 = pr.rowptr[(size_t)d];
      for (int64_t k = csr.rowptr[(size_t)d]; k < csr.rowptr[(size_t)d + 1]; ++k) {
        if (csr.val[(size_t)k] == 0.0) continue;
        if (k2 >= pr.rowptr[(size_t)d + 1] || pr.col[(size_t)k2] != csr.col[(size_t)k] || pr.val[(size_t)k2] != csr.val[(size_t)k]) ++pbad;
        ++k2;
      }
      if (k2 != pr.rowptr[(size_t)d + 1]) ++pbad;
      max_row = std::max<int64_t>(max_row, pr.rowptr[(size_t)d + 1] - pr.rowptr[(size_t)d]);
    }
    if (max_row != pr.max_row_nnz) ++pbad;
    printf("PRUNEBAD %lld %lld\n", pbad, (long long)dropped);
  }
  // launch grids beyond the limit: smm::split_batch cuts the (outer x inner) batch into parts that fit, every
  // batch row in exactly one part; a single row that cannot fit is reported, not launched
  {
    long long sbad = 0;
    struct Case { int64_t n_o, n_i, n_dblocks, n_lev, limit; };
    const Case cases[] = {{3600, 1, 254, 1, 0x7fffffffLL}, {3600, 1, 254, 1, 5000}, {120, 7, 1013, 75, 100000},
                          {1, 1000, 64, 3, 999}, {5, 5, 10, 1, 10}, {1, 1, 10, 1, 10}, {17, 3, 49152, 1, 49152}};
    for (const Case& c : cases) {
      // the tile kernel's rule (smm_launch.hpp): walks of 64 rows, halved while the grid stays below 4096
      auto blocks = [&](int64_t n_o, int64_t n_i) {
        const int64_t n_j = n_o * n_i;
        int64_t walk = 64;
        while (walk > 1 && c.n_dblocks * ((n_j + walk - 1) / walk) * c.n_lev < 4096) walk /= 2;
        walk = std::min(walk, n_j);
        return c.n_dblocks * ((n_j + walk - 1) / walk) * c.n_lev;
      };
      std::vector<int> seen((size_t)(c.n_o * c.n_i), 0);
      int64_t parts = 0;
      auto emit = [&](int64_t o0, int64_t n_o, int64_t i0, int64_t n_i) {
        if (blocks(n_o, n_i) > c.limit) ++sbad;
        for (int64_t o = o0; o < o0 + n_o; ++o)
          for (int64_t i = i0; i < i0 + n_i; ++i) ++seen[(size_t)(o * c.n_i + i)];
        ++parts;
        return 0;
      };
      const int rc = smm::split_batch(0, c.n_o, 0, c.n_i, c.limit, blocks, emit);
      const bool one_row_fits = blocks(1, 1) <= c.limit;
      if (rc != (one_row_fits ? 0 : -1)) ++sbad;
      if (one_row_fits)
        for (int v : seen) sbad += v != 1;
      if (c.limit == 0x7fffffffLL && parts != 1) ++sbad;
    }
    // an error of one part stops the walk and is passed on
    int calls = 0;
    if (smm::split_batch(0, 8, 0, 1, 1, [](int64_t a, int64_t b) { return a * b; },
                         [&](int64_t, int64_t, int64_t, int64_t) { return ++calls == 3 ? 7 : 0; }) != 7 || calls != 3)
      ++sbad;
    printf("SPLITBAD %lld\n", sbad);
  }
  // worker pools of the builders: a task that throws (std::bad_alloc from a worker's scratch vectors) must come
  // back as an exception on the calling thread after every started thread has been joined -- never
  // std::terminate --, and a thread that cannot be started costs parallelism, not the result
  {
    long long fbad = 0, thrown = 0;
    auto same_csr = [&](const smm::HostCsr& a) {
      return a.rowptr == csr.rowptr && a.col == csr.col && a.val == csr.val && a.n_used_src == csr.n_used_src;
    };
    smm::debug_builder_faults(true, -1);            // no thread can be started: every task runs on the caller
    {
      smm::HostCsr again;
      std::string e3;
      if (!smm::build_csr(n_src, n_dst, nnz, src.data(), dst.data(), w.data(), again, e3) || !same_csr(again)) ++fbad;
    }
    for (int no_threads = 0; no_threads < 2; ++no_threads) {
      for (int64_t at = 0; at < 40; ++at) {
        smm::debug_builder_faults(no_threads != 0, at);
        try {
          smm::HostCsr again;
          std::string e3;
          smm::HostSell sell2;
          smm::HostTilePlan plan2;
          if (!smm::build_csr(n_src, n_dst, nnz, src.data(), dst.data(), w.data(), again, e3)) ++fbad;
          smm::build_sell(again, sell2);
          smm::build_tile_plan(again, sell2, 256, 16, 512, plan2);
          if (!same_csr(again)) ++fbad;             // the fault counter ran past every task: a clean build
        } catch (const std::bad_alloc&) {
          ++thrown;
        }
      }
    }
    smm::debug_builder_faults(false, -1);
    smm::HostCsr again;
    std::string e3;
    if (!smm::build_csr(n_src, n_dst, nnz, src.data(), dst.data(), w.data(), again, e3) || !same_csr(again)) ++fbad;
    printf("FAULTBAD %lld %lld\n", fbad, thrown);
  }
  // the host pipelines' staging stages on the persistent pool (smm_hostpool.cpp): the pack gives the same block with
  // 1, 3 and 16 threads, plain and non-temporal stores, rows that do not fill a 16-row block; a worker that cannot
  // be started costs parallelism only; a task that throws comes back as a status (1 = bad_alloc) -- never
  // std::terminate -- and the pool is usable afterwards
  {
    long long qbad = 0, statuses = 0;
    const int64_t S = 5003, rows = 83, n_inner = 5, stride_i = S, stride_o = 2 * n_inner * S;   // a level of an (o, 2, 5, S) field
    const int64_t n_o = (rows + n_inner - 1) / n_inner;
    std::vector<double> x((size_t)(n_o * stride_o));
    for (size_t i = 0; i < x.size(); ++i) x[i] = 0.5 * (double)i;
    std::vector<int32_t> used;
    for (int32_t c = 3; c < S; c += 1 + (c % 7)) used.push_back(c);
    const int64_t U = (int64_t)used.size();
    auto entry = [&](int64_t r, int32_t c) { return x[(size_t)((r / n_inner) * stride_o + (r % n_inner) * stride_i + c)]; };
    auto check = [&](const std::vector<double>& out) {
      long long b2 = 0;
      for (int64_t u = 0; u < U; ++u)
        for (int64_t r = 0; r < rows; ++r) b2 += out[(size_t)(u * rows + r)] != entry(r, used[(size_t)u]);
      return b2;
    };
    for (int threads : {1, 3, 16})
      for (int streaming = 0; streaming < 2; ++streaming) {
        smm::set_host_threads(threads);
        std::vector<double> out((size_t)(U * rows), -1.0);
        if (smm::host_pack(out.data(), x.data(), 8, n_inner, stride_o, stride_i, used.data(), U, rows, streaming != 0)) ++qbad;
        qbad += check(out);
      }
    // f32, rows a multiple of 16 so that the streaming path is taken for every block
    {
      std::vector<float> xf(x.begin(), x.end()), outf((size_t)(U * 96), -1.f);
      std::vector<float> xrows((size_t)(96 * S));
      for (size_t i = 0; i < xrows.size(); ++i) xrows[i] = (float)(i % 100003);
      smm::set_host_threads(4);
      if (smm::host_pack(outf.data(), xrows.data(), 4, 96, 0, S, used.data(), U, 96, true)) ++qbad;
      for (int64_t u = 0; u < U; ++u)
        for (int64_t r = 0; r < 96; ++r) qbad += outf[(size_t)(u * 96 + r)] != xrows[(size_t)(r * S + used[(size_t)u])];
    }
    // big enough for the pool to be used: 1500 x 5003 doubles packed (> 4 MiB), and a 64-MiB copy
    const int64_t big_rows = 1504;
    std::vector<double> xb((size_t)(big_rows * S));
    for (size_t i = 0; i < xb.size(); ++i) xb[i] = (double)(i % 1000003);
    std::vector<double> outb((size_t)(U * big_rows));
    auto pack_big = [&]() { return smm::host_pack(outb.data(), xb.data(), 8, big_rows, 0, S, used.data(), U, big_rows, true); };
    auto big_ok = [&]() {
      long long b2 = 0;
      for (int64_t u = 0; u < U; u += 17)
        for (int64_t r = 0; r < big_rows; r += 5) b2 += outb[(size_t)(u * big_rows + r)] != xb[(size_t)(r * S + used[(size_t)u])];
      return b2;
    };
    smm::set_host_threads(6);
    if (pack_big() || big_ok()) ++qbad;
    const int workers_after_first = smm::pool_workers();
    if (pack_big() || smm::pool_workers() != workers_after_first) ++qbad;       // persistent: no new threads per call
    std::vector<char> ca((size_t)64 << 20), cb((size_t)64 << 20, 0);
    for (size_t i = 0; i < ca.size(); i += 4099) ca[i] = (char)(i * 31);
    if (smm::host_copy(cb.data(), ca.data(), ca.size()) || memcmp(ca.data(), cb.data(), ca.size()) != 0) ++qbad;
    // no worker can be started: the caller does everything, same block
    smm::debug_pool_faults(true, -1);
    std::fill(outb.begin(), outb.end(), -1.0);
    if (pack_big() || big_ok()) ++qbad;
    // a task throws: status 1, no terminate, with and without workers; then a clean run
    for (int no_threads = 0; no_threads < 2; ++no_threads)
      for (int64_t at = 0; at < 12; ++at) {
        smm::debug_pool_faults(no_threads != 0, at);
        const int rc = pack_big();
        if (rc == 1) ++statuses;
        else if (rc != 0) ++qbad;
      }
    smm::debug_pool_faults(false, -1);
    std::fill(outb.begin(), outb.end(), -1.0);
    if (pack_big() || big_ok()) ++qbad;
    smm::set_host_threads(argc > 1 ? atoi(argv[1]) : 0);
    printf("POOLBAD %lld %lld %d %d\n", qbad, statuses, workers_after_first, smm::usable_cpus());
  }
  printf("SELLBAD %lld\n", bad);
  return 0;
}
