// Practical HBM ceilings of MI355X for the access SHAPES of the regrid kernels (measurement tool, not
// part of the product): how fast can the chip move bytes when reads come in runs of a given length at
// scattered addresses and writes go out in segments of a given length on a given stride, at a given
// read : write mix?  The kernels' PMC traffic rates are held against these numbers in DESIGN.md.
//
//   hipcc --offload-arch=gfx950 -O3 tools/exp/ceiling.hip -o tools/exp/ceiling
//   tools/exp/ceiling <read_run_B> <read_B_per_unit> <write_seg_B> <write_B_per_unit> <write_stride_B> [units] [wg_per_cu] [reps]
//                     [read_pitch_B] [rows_per_tile] [slab_pitch_B] [exact_rounds] [barrier]
//
// One wave = one "unit" at a time: it reads read_B_per_unit bytes as 1-KiB wave loads (16 B per lane,
// eight in flight) whose bytes are cut into runs of read_run_B at pseudo-random 128-B-aligned places
// of an 8-GiB buffer (beyond the 256-MiB Infinity Cache), and writes write_B_per_unit bytes as 1-KiB
// wave stores (non-temporal) cut into segments of write_seg_B placed write_stride_B apart, neighbouring
// units writing neighbouring segments of the same rows (the Y tiles of the regrid kernels).
// With read_pitch_B > 0 the reads are not scattered but TILES of a row-major array of that pitch, as a tile plan
// lists them: unit u reads read_B_per_unit / read_run_B runs of read_run_B, one per array row, at the same column,
// and neighbouring units read neighbouring columns of the same rows (every byte still read once: no L2 reuse).
// rows_per_tile < runs per unit cuts a unit's runs into several tiles; with slab_pitch_B > 0 they are the SAME tile of
// consecutive slabs (batch rows) that far apart -- a workgroup walking the batch -- else consecutive tiles of one slab.
// (Units should read a multiple of 8 KiB: the load loop issues eight 1-KiB wave loads at a time and repeats the last
// piece to fill a short round, which costs issue slots without counting as bytes -- the round-3/4 "config-4 shape" line
// with its 2-KiB units measured that loop, not the chip.)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                         \
      return 1;                                                                       \
    }                                                                                 \
  } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Args {
  const char* src;
  char* dst;
  uint64_t slot_mask;      // number of run-sized slots in src (a power of two) - 1
  uint32_t slot_shift;     // log2 of the slot size (>= read_run, >= 128)
  uint32_t run_shift, seg_shift;   // log2(read_run), log2(write_seg): both powers of two, no division per piece
  uint32_t read_run, read_unit, write_seg, write_unit;
  uint64_t write_stride;
  uint64_t n_units;
  uint32_t cols;           // segments per row of the write pattern
  uint64_t read_pitch;     // > 0: reads are tiles of a row-major array of this pitch (bytes)
  uint32_t rcols;          // tiles per row band of that array
  uint64_t src_mask;       // buffer size - 1
  uint32_t rows_per_tile;  // runs of one tile (<= runs per unit)
  uint64_t slab_pitch;     // > 0: a unit's tiles are one tile position in consecutive slabs this far apart
  uint32_t exact_rounds;   // 0: rounds of eight loads (a short last round repeats its last piece); 2 / 4 / 8: loads per round
  uint32_t barrier;        // 1: the four waves of a workgroup meet at a barrier after every unit (a kernel staging per batch row)
};

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Q = loads a wave issues per round before it waits (8 by default; exact_rounds = units whose read bytes are a multiple
// of Q KiB issue exactly their own loads -- 2 or 4 for units of one or two batch rows of a small tile)
template <int Q>
__global__ __launch_bounds__(256) void ceiling_kernel(Args a) {
  const int lane = threadIdx.x & 63;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  const uint32_t n_rp = a.read_unit / 1024, n_wp = a.write_unit / 1024;
  const uint32_t runs_per_unit = (a.read_unit + a.read_run - 1) >> a.run_shift;
  const uint32_t segs_per_unit = (a.write_unit + a.write_seg - 1) >> a.seg_shift;
  u32x4 acc = {0, 0, 0, 0};
  for (uint64_t u = wave; u < a.n_units; u += n_waves) {
    for (uint32_t p0 = 0; p0 < n_rp; p0 += Q) {
      u32x4 v[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const uint32_t p = p0 + q < n_rp ? p0 + q : n_rp - 1;
        const uint32_t o = p * 1024 + lane * 16;
        const uint32_t run = o >> a.run_shift, within = o & (a.read_run - 1);
        uint64_t at;
        if (a.read_pitch) {
          const uint32_t k = run / a.rows_per_tile, row = run - k * a.rows_per_tile;   // k-th tile of the unit
          const uint32_t tiles_per_unit = (runs_per_unit + a.rows_per_tile - 1) / a.rows_per_tile;
          const uint64_t tile = a.slab_pitch ? u : u * tiles_per_unit + k;
          const uint64_t band = tile / a.rcols, col = tile - band * a.rcols;
          at = ((a.slab_pitch ? k * a.slab_pitch : 0) + (band * a.rows_per_tile + row) * a.read_pitch + col * a.read_run +
                within) & a.src_mask;
        } else {
          const uint64_t slot = mix(u * runs_per_unit + run + 0x9E3779B97F4A7C15ull) & a.slot_mask;
          at = (slot << a.slot_shift) + within;
        }
        v[q] = __builtin_nontemporal_load((const u32x4*)(a.src + at));
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) acc ^= v[q];
    }
    const uint64_t urow = u / a.cols, ucol = u - urow * a.cols;   // once per unit
    for (uint32_t p = 0; p < n_wp; ++p) {
      const uint32_t o = p * 1024 + lane * 16;
      const uint32_t seg = o >> a.seg_shift, within = o & (a.write_seg - 1);
      char* d = a.dst + (urow * segs_per_unit + seg) * a.write_stride + ucol * a.write_seg + within;
      u32x4 out = acc;
      out.x += p;
      __builtin_nontemporal_store(out, (u32x4*)d);
    }
    if (a.barrier) __syncthreads();
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u && n_wp == 0) a.dst[0] = 1;   // keeps the loads alive
}

int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s read_run_B read_B_per_unit write_seg_B write_B_per_unit write_stride_B [units] [wg_per_cu] [reps]\n", argv[0]);
    return 2;
  }
  Args a{};
  a.read_run = (uint32_t)atoll(argv[1]);
  a.read_unit = (uint32_t)atoll(argv[2]);
  a.write_seg = (uint32_t)atoll(argv[3]);
  a.write_unit = (uint32_t)atoll(argv[4]);
  a.write_stride = (uint64_t)atoll(argv[5]);
  a.n_units = argc > 6 ? (uint64_t)atoll(argv[6]) : 400000;
  const int wg_per_cu = argc > 7 ? atoi(argv[7]) : 8;
  const int reps = argc > 8 ? atoi(argv[8]) : 5;
  a.read_pitch = argc > 9 ? (uint64_t)atoll(argv[9]) : 0;
  a.rows_per_tile = argc > 10 ? (uint32_t)atoll(argv[10]) : 0;
  a.slab_pitch = argc > 11 ? (uint64_t)atoll(argv[11]) : 0;
  a.exact_rounds = argc > 12 ? (uint32_t)atoi(argv[12]) : 0;
  a.barrier = argc > 13 ? (uint32_t)atoi(argv[13]) : 0;
  auto pow2 = [](uint32_t v) { return v >= 16 && (v & (v - 1)) == 0; };
  if (a.read_unit % 1024 || a.write_unit % 1024 || !pow2(a.read_run) || !pow2(a.write_seg) ||
      a.write_stride < a.write_seg || a.write_stride % a.write_seg) {
    fprintf(stderr, "bytes per unit: multiples of 1024; runs / segments: powers of two >= 16; stride: a multiple of the segment\n");
    return 2;
  }
  const uint64_t src_bytes = 8ull << 30;
  auto lg = [](uint64_t v) { uint32_t s = 0; while ((1ull << s) < v) ++s; return s; };
  a.run_shift = lg(a.read_run);
  a.seg_shift = lg(a.write_seg);
  a.slot_shift = a.run_shift < 7 ? 7 : a.run_shift;
  a.slot_mask = (src_bytes >> a.slot_shift) - 1;
  a.src_mask = src_bytes - 1;
  if (a.read_pitch && (a.read_pitch % a.read_run || a.read_pitch % 16)) {
    fprintf(stderr, "read pitch: a multiple of the read run\n");
    return 2;
  }
  a.rcols = a.read_pitch ? (uint32_t)(a.read_pitch / a.read_run) : 1;
  {
    const uint32_t runs = (a.read_unit + a.read_run - 1) / a.read_run;
    if (a.rows_per_tile == 0 || a.rows_per_tile > runs) a.rows_per_tile = runs ? runs : 1;
  }
  a.cols = a.write_unit ? (uint32_t)(a.write_stride / a.write_seg) : 1;
  const uint32_t segs_per_unit = a.write_unit ? (a.write_unit + a.write_seg - 1) / a.write_seg : 0;
  const uint64_t rows = (a.n_units + a.cols - 1) / a.cols;
  const uint64_t dst_bytes = a.write_unit ? rows * segs_per_unit * a.write_stride + 4096 : 4096;
  if (dst_bytes > (200ull << 30)) {
    fprintf(stderr, "write pattern needs %.1f GB\n", dst_bytes / 1e9);
    return 2;
  }
  // placement sweep (argv[14] = largest offset in bytes): ONE allocation holds both buffers, the write buffer starts at
  // (end of the read buffer) + delta for a list of deltas -- does the rate follow the relative placement of the two streams?
  const uint64_t sweep = argc > 14 ? (uint64_t)atoll(argv[14]) : 0;
  char *src = nullptr, *dst = nullptr, *dst_base = nullptr;
  bool vmm_used = false;   // buffers from the virtual-memory API are left to process exit
  if (sweep) {
    CHECK(hipMalloc((void**)&src, src_bytes + dst_bytes + sweep + 4096));
    CHECK(hipMemset(src, 1, src_bytes + dst_bytes + sweep + 4096));
    dst = src + src_bytes;
  } else {
    // placement study (round 6): argv[15] = bytes by which the write buffer's base is moved inside its own over-sized
    // allocation (the library can do that for buffers it owns), argv[16] = bytes of a dummy allocation made BEFORE the two
    // buffers (moves where hipMalloc puts them).  The line printed below carries both base pointers, their difference modulo
    // 2 MiB / 64 MiB / 1 GiB and the allocation ranges, so that a fast placement can be told from a slow one.
    const uint64_t dst_off = argc > 15 ? (uint64_t)atoll(argv[15]) : 0;
    const uint64_t pre = argc > 16 ? (uint64_t)atoll(argv[16]) : 0;
    char* dummy = nullptr;
    if (pre) CHECK(hipMalloc((void**)&dummy, pre));
    // argv[18] = alignment (bytes) of a virtual-memory-API allocation for the write buffer instead of hipMalloc
    // (hipMemCreate + hipMemAddressReserve(alignment) + hipMemMap): does a virtual address aligned to 64 MiB / 1 GiB
    // let the driver map larger fragments?  argv[19] != 0: the read buffer the same way.
    const uint64_t vmm_align = argc > 18 ? (uint64_t)atoll(argv[18]) : 0;
    const int vmm_src = argc > 19 ? atoi(argv[19]) : 0;
    auto vmm_alloc = [&](char** out, uint64_t bytes) -> int {
      hipMemAllocationProp prop = {};
      prop.type = hipMemAllocationTypePinned;
      prop.location.type = hipMemLocationTypeDevice;
      prop.location.id = 0;
      size_t gran = 0;
      CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
      const uint64_t unit = vmm_align > gran ? vmm_align : gran;
      const uint64_t size = (bytes + unit - 1) / unit * unit;
      hipMemGenericAllocationHandle_t h;
      CHECK(hipMemCreate(&h, size, &prop, 0));
      void* p = nullptr;
      CHECK(hipMemAddressReserve(&p, size, vmm_align, nullptr, 0));
      CHECK(hipMemMap(p, size, 0, h, 0));
      hipMemAccessDesc ad = {};
      ad.location = prop.location;
      ad.flags = hipMemAccessFlagsProtReadWrite;
      CHECK(hipMemSetAccess(p, size, &ad, 1));
      *out = (char*)p;
      fprintf(stderr, "vmm: %llu bytes at %p (granularity %zu, alignment %llu)\n", (unsigned long long)size, p, gran, (unsigned long long)vmm_align);
      return 0;
    };
    if (vmm_align && vmm_src) {
      if (vmm_alloc(&src, src_bytes)) return 1;
    } else {
      CHECK(hipMalloc((void**)&src, src_bytes));
    }
    if (vmm_align) {
      if (vmm_alloc(&dst, dst_bytes + dst_off)) return 1;
      vmm_used = true;
    } else {
      CHECK(hipMalloc((void**)&dst, dst_bytes + dst_off));
    }
    CHECK(hipMemset(src, 1, src_bytes));
    CHECK(hipMemset(dst, 0, dst_bytes + dst_off));
    dst_base = dst;
    dst += dst_off;
  }
  a.src = src;
  a.dst = dst;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const unsigned grid = (unsigned)(prop.multiProcessorCount * wg_per_cu);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int q_round = a.exact_rounds ? (int)a.exact_rounds : 8;
  if ((q_round != 2 && q_round != 4 && q_round != 8) || (a.exact_rounds && (a.read_unit / 1024) % q_round)) {
    fprintf(stderr, "exact_rounds: 2, 4 or 8 loads per round, dividing the unit's KiB\n");
    return 2;
  }
  auto launch = [&]() {
    if (q_round == 2) hipLaunchKernelGGL(ceiling_kernel<2>, dim3(grid), dim3(256), 0, nullptr, a);
    else if (q_round == 4) hipLaunchKernelGGL(ceiling_kernel<4>, dim3(grid), dim3(256), 0, nullptr, a);
    else hipLaunchKernelGGL(ceiling_kernel<8>, dim3(grid), dim3(256), 0, nullptr, a);
  };
  launch();
  CHECK(hipDeviceSynchronize());
  if (sweep) {
    const double bytes = (double)a.n_units * a.read_unit + (double)a.n_units * a.write_unit;
    printf("src %p\n", (void*)src);
    for (uint64_t delta = 0; delta <= sweep; delta = delta ? delta * 2 : 256) {
      a.dst = src + src_bytes + delta;
      launch();
      CHECK(hipDeviceSynchronize());
      float sum2 = 0.f;
      for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0, nullptr));
        launch();
        CHECK(hipEventRecord(e1, nullptr));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        sum2 += ms;
      }
      printf("delta %12llu  %.0f GB/s\n", (unsigned long long)delta, bytes / (sum2 / reps) / 1e6);
    }
    CHECK(hipFree(src));
    return 0;
  }
  // placement study, second question (argv[17] = pairs): is the rate a property of the PROCESS or of the ALLOCATION?
  // The first pair of buffers is kept and measured again after every further pair (allocated while all earlier ones
  // are still held, so that each lands in other physical memory) has been measured.
  const int more_pairs = argc > 17 ? atoi(argv[17]) : 0;
  if (more_pairs > 0) {
    const double bytes = (double)a.n_units * a.read_unit + (double)a.n_units * a.write_unit;
    auto measure = [&](const char* s0, char* d0, float* per_rep) -> double {
      a.src = s0;
      a.dst = d0;
      launch();
      (void)hipDeviceSynchronize();
      float sum3 = 0.f;
      for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(e0, nullptr);
        launch();
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (per_rep) per_rep[r] = (float)(bytes / ms / 1e6);
        sum3 += ms;
      }
      return bytes / (sum3 / reps) / 1e6;
    };
    std::vector<float> per((size_t)reps);
    printf("pair 0 (src %p dst %p): %.0f GB/s  reps:", (void*)src, (void*)dst, measure(src, dst, per.data()));
    for (float v : per) printf(" %.0f", v);
    printf("\n");
    std::vector<char*> held;
    for (int k = 1; k <= more_pairs; ++k) {
      char *s2 = nullptr, *d2 = nullptr;
      if (hipMalloc((void**)&s2, src_bytes) != hipSuccess || hipMalloc((void**)&d2, dst_bytes) != hipSuccess) {
        printf("pair %d: out of device memory\n", k);
        break;
      }
      (void)hipMemset(s2, 1, src_bytes);
      (void)hipMemset(d2, 0, dst_bytes);
      held.push_back(s2);
      held.push_back(d2);
      const double r2 = measure(s2, d2, nullptr);
      const double r0 = measure(src, dst, nullptr);
      // crossed: reads from the new pair's source, writes into the first pair's destination, and the other way round
      const double rx = measure(s2, dst, nullptr), ry = measure(src, d2, nullptr);
      printf("pair %d (src %p dst %p): %.0f GB/s   pair 0 again: %.0f   src%d->dst0: %.0f   src0->dst%d: %.0f\n", k, (void*)s2, (void*)d2, r2, r0, k, rx, k, ry);
    }
    for (char* h : held) (void)hipFree(h);
    CHECK(hipFree(src));
    CHECK(hipFree(dst_base));
    return 0;
  }
  float best = 1e30f, sum = 0.f;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0, nullptr));
    launch();
    CHECK(hipEventRecord(e1, nullptr));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double rb = (double)a.n_units * a.read_unit, wb = (double)a.n_units * a.write_unit;
  printf("{\"read_run\": %u, \"read_unit\": %u, \"write_seg\": %u, \"write_unit\": %u, \"write_stride\": %llu, "
         "\"read_pitch\": %llu, \"rows_per_tile\": %u, \"slab_pitch\": %llu, \"exact_rounds\": %u, \"barrier\": %u, \"units\": %llu, \"wg_per_cu\": %d, \"ms_mean\": %.4f, \"ms_best\": %.4f, \"read_GBs\": %.1f, "
         "\"write_GBs\": %.1f, \"total_GBs\": %.1f, \"write_share\": %.3f}\n",
         a.read_run, a.read_unit, a.write_seg, a.write_unit, (unsigned long long)a.write_stride,
         (unsigned long long)a.read_pitch, a.rows_per_tile, (unsigned long long)a.slab_pitch, a.exact_rounds, a.barrier, (unsigned long long)a.n_units, wg_per_cu, sum / reps, best, rb / (sum / reps) / 1e6, wb / (sum / reps) / 1e6,
         (rb + wb) / (sum / reps) / 1e6, wb / (rb + wb + 1e-30));
  {
    void *sb = nullptr, *db = nullptr;
    size_t ss = 0, ds = 0;
    (void)hipMemGetAddressRange((hipDeviceptr_t*)&sb, &ss, (hipDeviceptr_t)src);
    (void)hipMemGetAddressRange((hipDeviceptr_t*)&db, &ds, (hipDeviceptr_t)dst_base);
    const uint64_t ps = (uint64_t)(uintptr_t)src, pd = (uint64_t)(uintptr_t)dst, diff = pd > ps ? pd - ps : ps - pd;
    printf("placement {\"total_GBs\": %.1f, \"src\": \"0x%llx\", \"dst\": \"0x%llx\", \"dst_above_src\": %d, \"diff_mod_2M\": %llu, "
           "\"diff_mod_64M\": %llu, \"diff_mod_1G\": %llu, \"src_mod_1G\": %llu, \"dst_mod_1G\": %llu, \"src_mod_2M\": %llu, \"dst_mod_2M\": %llu, "
           "\"src_range\": [\"0x%llx\", %zu], \"dst_range\": [\"0x%llx\", %zu]}\n",
           (rb + wb) / (sum / reps) / 1e6, (unsigned long long)ps, (unsigned long long)pd, (int)(pd > ps),
           (unsigned long long)(diff % (2ull << 20)), (unsigned long long)(diff % (64ull << 20)), (unsigned long long)(diff % (1ull << 30)),
           (unsigned long long)(ps % (1ull << 30)), (unsigned long long)(pd % (1ull << 30)), (unsigned long long)(ps % (2ull << 20)),
           (unsigned long long)(pd % (2ull << 20)), (unsigned long long)(uintptr_t)sb, ss, (unsigned long long)(uintptr_t)db, ds);
  }
  if (!vmm_used) {
    CHECK(hipFree(src));
    CHECK(hipFree(dst_base));
  }
  return 0;
}
