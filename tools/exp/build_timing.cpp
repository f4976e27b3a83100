// Host-only timing of the operator builder (smm_build.cpp) on synthetic link lists of BASELINE sizes.
//   g++ -O3 -std=c++17 -pthread tools/exp/build_timing.cpp smmregrid_amd/csrc/smm_build.cpp -o /tmp/build_timing
//   /tmp/build_timing <n_dst> <links_per_row> <n_src> [shuffle] [threads]
// Reports ms per stage; with `shuffle` the links arrive in random order (the general path).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#include "../../smmregrid_amd/csrc/smm_internal.h"

static double now() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  const long long n_dst = argc > 1 ? atoll(argv[1]) : 12582912, per = argc > 2 ? atoll(argv[2]) : 4;
  const long long n_src = argc > 3 ? atoll(argv[3]) : 13107200;
  const bool shuffle = argc > 4 && atoi(argv[4]) != 0;
  if (argc > 5) smm::set_host_threads(atoi(argv[5]));
  const long long nnz = n_dst * per;
  std::vector<int32_t> src((size_t)nnz), dst((size_t)nnz);
  std::vector<double> w((size_t)nnz);
  std::mt19937_64 rng(7);
  for (long long d = 0; d < n_dst; ++d) {
    const long long base = (long long)((double)d / n_dst * (n_src - 6000));
    for (long long k = 0; k < per; ++k) {
      src[(size_t)(d * per + k)] = (int32_t)(per > 8 ? ((d / 360) * 5 + k / 6) * 1442 + (d % 360) * 4 + (k % 6) + 1 : base + (k / 2) * 5120 + (k % 2) + 1);
      dst[(size_t)(d * per + k)] = (int32_t)(d + 1);
      w[(size_t)(d * per + k)] = 1.0 / per;
    }
  }
  if (shuffle) {
    std::vector<int64_t> p((size_t)nnz);
    std::iota(p.begin(), p.end(), 0);
    std::shuffle(p.begin(), p.end(), rng);
    std::vector<int32_t> s2((size_t)nnz), d2((size_t)nnz);
    for (long long k = 0; k < nnz; ++k) { s2[(size_t)k] = src[(size_t)p[(size_t)k]]; d2[(size_t)k] = dst[(size_t)p[(size_t)k]]; }
    src.swap(s2); dst.swap(d2);
  }
  smm::HostCsr csr;
  std::string err;
  double t0 = now();
  if (!smm::build_csr(n_src, n_dst, nnz, src.data(), dst.data(), w.data(), csr, err)) { printf("ERR %s\n", err.c_str()); return 1; }
  double t1 = now();
  smm::HostSell sell;
  smm::build_sell(csr, sell);
  double t2 = now();
  smm::HostTilePlan plan;
  const int rows = csr.max_row_nnz > 16 ? 64 : 256;
  const int64_t budget = rows == 256 ? 2048 : 512;
  smm::build_tile_plan(csr, sell, rows, 4, budget, plan);
  double t3 = now();
  smm::tighten_tile_plan(csr, plan, budget);
  double t4 = now();
  unsigned long long h = 1469598103934665603ull;
  for (int32_t c : csr.col) h = (h ^ (unsigned)c) * 1099511628211ull;
  for (int32_t c : plan.chunk_src) h = (h ^ (unsigned)c) * 1099511628211ull;
  for (int32_t c : plan.lcol) h = (h ^ (unsigned)c) * 1099511628211ull;
  printf("links %lld: build_csr %.0f ms, build_sell %.0f ms, build_tile_plan %.0f ms, tighten %.0f ms, total %.0f ms"
         "  (nnz %lld U %lld max_row %lld chunks %lld lines %lld distinct %lld valid %d hash %llx)\n",
         nnz, t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0, (long long)csr.nnz, (long long)csr.n_used_src,
         (long long)csr.max_row_nnz, (long long)plan.total_chunks, (long long)plan.total_lines,
         (long long)plan.distinct_chunks, (int)plan.valid, h);
  return 0;
}
