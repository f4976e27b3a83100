// Host side of the host-buffer pipelines (smm_apply_host / smm_group_apply_host): the staging
// copies that feed the pinned buffers -- the plain parallel copy and the pack of the used source cells --
// on ONE persistent worker pool per process.  Host only (no HIP): tests/cpp/build_harness.cpp links it.
//
// Nothing in here throws to its caller: the extern "C" entries above must return a status
// (include/smmregrid_amd.h).  Tasks are claimed one by one from a shared counter by the calling thread and by
// whatever workers exist, so a worker that cannot be started (EAGAIN under a pids limit) costs parallelism,
// never the result; whatever a task body throws is caught on its thread and comes back as a code.
#include "smm_internal.h"

#include <emmintrin.h>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <thread>

namespace smm {

// ---------------------------------------------------------------------------------------- usable CPUs

namespace {

// tightest CPU quota (quota / period) of `file` in `dir` and every ancestor up to `root`; 0 = none found
double quota_up(const std::string& root, std::string rel, bool v2) {
  double best = 0.0;
  auto take = [&](double q, double period) {
    if (q > 0 && period > 0) best = best == 0.0 ? q / period : std::min(best, q / period);
  };
  while (!rel.empty() && rel.back() == '/') rel.pop_back();
  for (;;) {
    const std::string dir = root + rel;
    if (v2) {
      if (FILE* f = fopen((dir + "/cpu.max").c_str(), "r")) {
        char q[64] = {0};
        double period = 0;
        if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0) take(atof(q), period);
        fclose(f);
      }
    } else {
      double q = 0, period = 0;
      if (FILE* f = fopen((dir + "/cpu.cfs_quota_us").c_str(), "r")) {
        if (fscanf(f, "%lf", &q) != 1) q = 0;
        fclose(f);
      }
      if (FILE* f = fopen((dir + "/cpu.cfs_period_us").c_str(), "r")) {
        if (fscanf(f, "%lf", &period) != 1) period = 0;
        fclose(f);
      }
      take(q, period);
    }
    if (rel.empty()) break;
    const size_t cut = rel.find_last_of('/');
    rel = cut == std::string::npos ? std::string() : rel.substr(0, cut);
  }
  return best;
}

int compute_usable_cpus() {
  int affinity = 0;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) affinity = CPU_COUNT(&set);
  if (affinity <= 0) affinity = (int)std::max<unsigned>(1, std::thread::hardware_concurrency());
  double quota = 0.0;
  try {
    std::string rel_v2, rel_v1;
    bool have_v2 = false, have_v1 = false;
    if (FILE* f = fopen("/proc/self/cgroup", "r")) {
      char line[4096];
      while (fgets(line, sizeof(line), f)) {
        std::string s(line);
        while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
        const size_t a = s.find(':'), b = a == std::string::npos ? a : s.find(':', a + 1);
        if (b == std::string::npos) continue;
        const std::string id = s.substr(0, a), ctl = s.substr(a + 1, b - a - 1), path = s.substr(b + 1);
        if (id == "0" && ctl.empty()) {
          rel_v2 = path;
          have_v2 = true;
        } else if ((',' + ctl + ',').find(",cpu,") != std::string::npos) {
          rel_v1 = path;
          have_v1 = true;
        }
      }
      fclose(f);
    }
    auto tighter = [&](double q) {
      if (q > 0) quota = quota == 0.0 ? q : std::min(quota, q);
    };
    tighter(quota_up("/sys/fs/cgroup", have_v2 ? rel_v2 : std::string(), true));
    for (const char* root : {"/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"})
      tighter(quota_up(root, have_v1 ? rel_v1 : std::string(), false));
  } catch (...) {   // out of memory while reading a path: the affinity count stands
  }
  if (quota > 0) affinity = std::max(1, std::min(affinity, (int)(quota + 0.5)));
  return affinity;
}

// CPU sets of the NUMA nodes this process may run on (each intersected with the process's affinity mask; nodes without an
// allowed CPU are left out).  Worker k of the staging pool confines itself to node k % n: the pinned staging buffers live on the
// GPU's node and the caller's arrays wherever they were first touched, so the pack's threads should sit on BOTH -- left to the
// scheduler they sometimes all land on the node far from the pinned memory and the pack runs at half speed (profiles/
// r06_host_to_host.txt: 25.7 against 10.6 - 13.4 ms per 512 rows with the threads on the far node / the near node / both).
std::vector<cpu_set_t> numa_node_sets() {
  std::vector<cpu_set_t> nodes;
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return nodes;
  for (int node = 0; node < 64; ++node) {
    char path[96];
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) {
      if (node > 8) break;   // node numbers may have holes, but not long ones
      continue;
    }
    char line[4096] = {0};
    const bool got = fgets(line, sizeof(line), f) != nullptr;
    fclose(f);
    if (!got) continue;
    cpu_set_t set;
    CPU_ZERO(&set);
    int n_in = 0;
    for (char* p = line; *p;) {                         // "0-63,128-191"
      char* end = nullptr;
      long a = strtol(p, &end, 10);
      if (end == p) break;
      long b = a;
      if (*end == '-') {
        p = end + 1;
        b = strtol(p, &end, 10);
      }
      for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
        if (c >= 0 && CPU_ISSET((int)c, &allowed)) {
          CPU_SET((int)c, &set);
          ++n_in;
        }
      p = (*end == ',') ? end + 1 : end;
      if (*end != ',' ) break;
    }
    if (n_in > 0) nodes.push_back(set);
  }
  return nodes;
}

std::atomic<bool> g_pool_no_threads{false};
std::atomic<int64_t> g_pool_throw_in_task{-1};

// ---------------------------------------------------------------------------------------- the pool

struct Job {
  void (*fn)(void*, int64_t) = nullptr;
  void* ctx = nullptr;
  int64_t n = 0;
  int helpers = 0;                    // workers 0 .. helpers-1 take part
  std::atomic<int64_t> next{0};
  std::atomic<int> err{0};            // first failure: 1 = bad_alloc, 2 = anything else
};

void claim(Job& j) noexcept {
  for (;;) {
    const int64_t i = j.next.fetch_add(1, std::memory_order_relaxed);
    if (i >= j.n) return;
    int code = 0;
    try {
      if (g_pool_throw_in_task.load(std::memory_order_relaxed) >= 0 && g_pool_throw_in_task.fetch_sub(1) == 0)
        throw std::bad_alloc();
      j.fn(j.ctx, i);
    } catch (const std::bad_alloc&) {
      code = 1;
    } catch (...) {
      code = 2;
    }
    if (code) {
      int expected = 0;
      j.err.compare_exchange_strong(expected, code);
      j.next.store(j.n, std::memory_order_relaxed);   // the other threads stop claiming
      return;
    }
  }
}

class Pool {
 public:
  int run(int64_t n_tasks, int max_threads, void (*fn)(void*, int64_t), void* ctx) noexcept {
    if (n_tasks <= 0) return 0;
    Job job;
    job.fn = fn;
    job.ctx = ctx;
    job.n = n_tasks;
    const int want = (int)std::min<int64_t>(std::max(1, max_threads), n_tasks) - 1;   // helpers besides the caller
    if (want > 0) {
      std::unique_lock<std::mutex> one_job(job_mu_);     // jobs of concurrent callers take turns (each lasts ms)
      {
        std::unique_lock<std::mutex> lk(mu_);
        while (!stop_ && (int)workers_.size() < want && !g_pool_no_threads.load(std::memory_order_relaxed)) {
          try {
            workers_.emplace_back(&Pool::worker, this, (int)workers_.size(), gen_);
          } catch (...) {   // EAGAIN / out of memory: go on with the workers there are
            break;
          }
        }
        // after shutdown() (process exit has begun; a straggling thread may still call in) the caller works alone
        job.helpers = (stop_ || g_pool_no_threads.load(std::memory_order_relaxed)) ? 0 : std::min(want, (int)workers_.size());
        if (job.helpers > 0) {
          job_ = &job;
          busy_ = job.helpers;
          ++gen_;
        }
      }
      if (job.helpers > 0) cv_work_.notify_all();
      claim(job);
      if (job.helpers > 0) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [&] { return busy_ == 0; });
        job_ = nullptr;
      }
    } else {
      claim(job);
    }
    return job.err.load();
  }

  int size() {
    std::unique_lock<std::mutex> lk(mu_);
    return (int)workers_.size();
  }

  void shutdown() noexcept {
    {
      std::unique_lock<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_work_.notify_all();
    for (auto& th : workers_)
      if (th.joinable()) th.join();
    workers_.clear();
  }

 private:
  void worker(int index, uint64_t seen) noexcept {
    try {
      static const std::vector<cpu_set_t> nodes = numa_node_sets();
      if (nodes.size() > 1) {
        const cpu_set_t& mine = nodes[(size_t)index % nodes.size()];
        (void)sched_setaffinity(0, sizeof(mine), &mine);   // a refusal changes nothing: the thread stays where it may run
      }
    } catch (...) {
    }
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_work_.wait(lk, [&] { return stop_ || gen_ != seen; });
      if (stop_) return;
      seen = gen_;
      Job* j = job_;
      if (!j || index >= j->helpers) continue;
      lk.unlock();
      claim(*j);
      lk.lock();
      if (--busy_ == 0) cv_done_.notify_all();
    }
  }

  std::mutex job_mu_, mu_;
  std::condition_variable cv_work_, cv_done_;
  std::vector<std::thread> workers_;
  Job* job_ = nullptr;
  uint64_t gen_ = 0;
  int busy_ = 0;
  bool stop_ = false;
};

std::atomic<Pool*> g_pool{nullptr};
std::atomic<int> g_pool_pid{0};

// the pool of THIS process: a child of fork() has none of the parent's threads and gets a pool of its own
// (the parent's object is left alone -- its mutexes may have been held at the fork)
Pool* pool() noexcept {
  const int pid = (int)getpid();
  Pool* p = g_pool.load(std::memory_order_acquire);
  if (p && g_pool_pid.load(std::memory_order_acquire) == pid) return p;
  Pool* fresh = new (std::nothrow) Pool();
  if (!fresh) return nullptr;
  static std::mutex create_mu;
  std::lock_guard<std::mutex> lock(create_mu);
  p = g_pool.load(std::memory_order_acquire);
  if (p && g_pool_pid.load(std::memory_order_acquire) == pid) {
    delete fresh;
    return p;
  }
  g_pool.store(fresh, std::memory_order_release);
  g_pool_pid.store(pid, std::memory_order_release);
  return fresh;
}

struct Reaper {   // at exit / dlclose: the workers are stopped and joined (no thread outlives the library's code)
  ~Reaper() {
    Pool* p = g_pool.load(std::memory_order_acquire);
    if (p && g_pool_pid.load(std::memory_order_acquire) == (int)getpid()) p->shutdown();
  }
} g_reaper;

}  // namespace

int usable_cpus() {
  static const int n = compute_usable_cpus();
  return n;
}

void debug_pool_faults(bool no_threads, int64_t throw_in_task) {
  g_pool_no_threads.store(no_threads);
  g_pool_throw_in_task.store(throw_in_task);
}

int pool_workers() {
  Pool* p = g_pool.load(std::memory_order_acquire);
  return p && g_pool_pid.load(std::memory_order_acquire) == (int)getpid() ? p->size() : 0;
}

int pool_run(int64_t n_tasks, int max_threads, void (*fn)(void*, int64_t), void* ctx) noexcept {
  Pool* p = max_threads > 1 && n_tasks > 1 ? pool() : nullptr;
  if (!p) {   // single-threaded by request, or not even the pool object could be allocated
    Job job;
    job.fn = fn;
    job.ctx = ctx;
    job.n = n_tasks;
    claim(job);
    return job.err.load();
  }
  return p->run(n_tasks, max_threads, fn, ctx);
}

// ---------------------------------------------------------------------------------------- staging copies

int staging_threads() {
  const int fixed = fixed_host_threads();
  return fixed > 0 ? fixed : std::min(16, usable_cpus());
}

namespace {
struct CopyJob {
  char* dst;
  const char* src;
  size_t bytes, per;
};
}  // namespace

int host_copy(void* dst, const void* src, size_t bytes) noexcept {
  const size_t kMin = 4u << 20;    // below this per thread a memcpy is not worth a wake-up
  int nt = (int)std::min<size_t>((size_t)std::min(8, staging_threads()), std::max<size_t>(1, bytes / kMin));
  if (nt <= 1) {
    memcpy(dst, src, bytes);
    return 0;
  }
  CopyJob job{(char*)dst, (const char*)src, bytes, ((bytes / ((size_t)nt * 4) + 4095) & ~(size_t)4095)};
  const int64_t n_tasks = (int64_t)((bytes + job.per - 1) / job.per);
  return pool_run(n_tasks, nt, [](void* c, int64_t i) {
    const CopyJob& j = *(const CopyJob*)c;
    const size_t lo = (size_t)i * j.per, hi = std::min(j.bytes, lo + j.per);
    memcpy(j.dst + lo, j.src + lo, hi - lo);
  }, &job);
}

namespace {
struct Copy2dJob {
  char* dst;
  const char* src;
  size_t dpitch, spitch, width;
  int64_t height, per;
};
}  // namespace

int host_copy_2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, int64_t height) noexcept {
  if (height <= 0 || width == 0) return 0;
  const size_t total = width * (size_t)height;
  int nt = (int)std::min<size_t>((size_t)std::min(8, staging_threads()), std::max<size_t>(1, total / (4u << 20)));
  nt = (int)std::min<int64_t>(nt, height);
  Copy2dJob job{(char*)dst, (const char*)src, dpitch, spitch, width, height, 1};
  job.per = std::max<int64_t>(1, height / ((int64_t)std::max(nt, 1) * 4));
  return pool_run((height + job.per - 1) / job.per, nt, [](void* c, int64_t i) {
    const Copy2dJob& j = *(const Copy2dJob*)c;
    const int64_t r0 = i * j.per, r1 = std::min(j.height, r0 + j.per);
    for (int64_t r = r0; r < r1; ++r) memcpy(j.dst + (size_t)r * j.dpitch, j.src + (size_t)r * j.spitch, j.width);
  }, &job);
}

namespace {

// 16 batch entries of one used cell are 128 B (f64) / 64 B (f32) of the packed block: whole cache lines of a buffer
// nothing reads before the DMA engine does.  Written with non-temporal stores they do not pull the line in first
// (a plain store reads it for ownership: a third more host-memory traffic on a path that host memory bounds).
inline void store_run(double* dst, const double* const* row, int32_t c) {
  for (int k = 0; k < 16; k += 2) _mm_stream_pd(dst + k, _mm_set_pd(row[k + 1][c], row[k][c]));
}
inline void store_run(float* dst, const float* const* row, int32_t c) {
  for (int k = 0; k < 16; k += 4) _mm_stream_ps(dst + k, _mm_set_ps(row[k + 3][c], row[k + 2][c], row[k + 1][c], row[k][c]));
}

template <typename T>
void pack_rows_t(T* __restrict__ out, const T* __restrict__ x, int64_t n_inner, int64_t stride_o, int64_t stride_i,
                 const int32_t* __restrict__ used, int64_t u0, int64_t u1, int64_t rows, bool streaming) {
  constexpr int64_t RB = 16;
  for (int64_t r0 = 0; r0 < rows; r0 += RB) {
    const int64_t rn = std::min(RB, rows - r0);
    const T* row[RB];
    for (int64_t r = 0; r < rn; ++r)
      row[r] = x + (size_t)((r0 + r) / n_inner) * stride_o + (size_t)((r0 + r) % n_inner) * stride_i;
    const bool nt = streaming && rn == RB && ((uintptr_t)(out + (size_t)u0 * rows + r0) & 15) == 0 &&
                    (((size_t)rows * sizeof(T)) & 15) == 0;
    if (nt) {
      for (int64_t u = u0; u < u1; ++u) store_run(out + (size_t)u * rows + r0, row, used[u]);
    } else {
      for (int64_t u = u0; u < u1; ++u) {
        const int32_t c = used[u];
        T* dst = out + (size_t)u * rows + r0;
        for (int64_t r = 0; r < rn; ++r) dst[r] = row[r][c];
      }
    }
  }
  if (streaming) _mm_sfence();
}

struct PackJob {
  void* out;
  const void* x;
  size_t xsz;
  int64_t n_inner, stride_o, stride_i;
  const int32_t* used;
  int64_t U, rows, per;
  bool streaming;
};

void pack_task(void* c, int64_t i) {
  const PackJob& j = *(const PackJob*)c;
  const int64_t u0 = std::min(j.U, i * j.per), u1 = std::min(j.U, u0 + j.per);
  if (j.xsz == 8)
    pack_rows_t((double*)j.out, (const double*)j.x, j.n_inner, j.stride_o, j.stride_i, j.used, u0, u1, j.rows, j.streaming);
  else
    pack_rows_t((float*)j.out, (const float*)j.x, j.n_inner, j.stride_o, j.stride_i, j.used, u0, u1, j.rows, j.streaming);
}

}  // namespace

int host_pack(void* out, const void* x, size_t xsz, int64_t n_inner, int64_t stride_o, int64_t stride_i,
              const int32_t* used, int64_t U, int64_t rows, bool streaming) noexcept {
  if (U <= 0 || rows <= 0) return 0;
  int nt = staging_threads();
  if ((size_t)U * (size_t)rows * xsz < (4u << 20)) nt = 1;
  // ranges of used cells, four per thread so that a thread the scheduler took away does not hold up the rest
  PackJob job{out, x, xsz, std::max<int64_t>(n_inner, 1), stride_o, stride_i, used, U, rows, 0, streaming};
  const int64_t pieces = nt == 1 ? 1 : (int64_t)nt * 4;
  job.per = std::max<int64_t>(64, (U + pieces - 1) / pieces);
  return pool_run((U + job.per - 1) / job.per, nt, pack_task, &job);
}

}  // namespace smm
