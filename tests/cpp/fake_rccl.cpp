// TEST INFRASTRUCTURE -- never part of the product path.
//
// A stand-in for librccl that lets the world-size > 1 data path of smmregrid_amd
// (csrc/smm_comm.cpp -> smmregrid_amd/comm.py -> distributed.TiledRingGather -> bench.py's
// gather phase) execute on a box with ONE GPU: several rank processes share the device and
// `SMM_RCCL_LIB` names this library instead of librccl.so (smm_comm.cpp binds whatever that
// variable names).  It exports exactly the six symbols smm_comm.cpp resolves, with RCCL's own
// signatures (checked against <rccl/rccl.h> at compile time), implemented BLOCKING and
// HOST-STAGED:
//
//   ncclGather / ncclAllGather = hipStreamSynchronize(stream); D2H of the send buffer into a POSIX
//   shared-memory segment named by the unique id; a host barrier; the receiving ranks H2D from the
//   segment into recv + rank * count; a second barrier before the segment is reused.
//
// It moves the right bytes to the right places in the order RCCL documents (rank i's data at
// offset i * sendcount) and nothing else: no xGMI, no overlap, no performance meaning.  A rank that
// does not reach a barrier within SMM_FAKE_RCCL_TIMEOUT_S (default 60) makes every waiting rank
// print a line to stderr and _exit(86): a broken schedule fails a test instead of hanging a box.
//
// Build (tests do it):  g++ -O1 -shared -fPIC -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include
//                       fake_rccl.cpp -o libfake_rccl.so -L/opt/rocm/lib -lamdhip64 -lrt
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

constexpr uint64_t kMagic = 0x736d6d66616b6572ull;   // "smmfaker"

struct Header {                       // lives at the start of the segment; a fresh segment is all zeros
  std::atomic<uint64_t> magic;
  std::atomic<int> attached;          // ranks that mapped the segment (ncclCommInitRank blocks on it)
  std::atomic<int> arrived;           // barrier: arrivals of the current generation
  std::atomic<int> generation;        // barrier: bumped by the last arrival
  std::atomic<int> detached;
  std::atomic<long long> collectives; // calls completed (rank 0 counts)
  char pad[64];
};

struct FakeComm {
  Header* hdr = nullptr;
  unsigned char* data = nullptr;      // n_ranks slots of slot_bytes each
  size_t map_bytes = 0, slot_bytes = 0;
  int n_ranks = 0, rank = 0;
  char name[64] = {0};
};

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double timeout_s() {
  const char* e = getenv("SMM_FAKE_RCCL_TIMEOUT_S");
  double t = e ? atof(e) : 60.0;
  return t > 0 ? t : 60.0;
}

size_t slot_bytes_from_env() {
  const char* e = getenv("SMM_FAKE_RCCL_SLOT_BYTES");   // tests set a tiny slot to exercise the chunk loop
  long long v = e ? atoll(e) : (64ll << 20);
  if (v < 64) v = 64;
  return (size_t)(v & ~63ll);
}

[[noreturn]] void die(const FakeComm* c, const char* what) {
  fprintf(stderr, "fake_rccl: rank %d of %d: %s -- giving up (exit 86)\n", c ? c->rank : -1, c ? c->n_ranks : -1, what);
  fflush(stderr);
  _exit(86);
}

void wait_until(const FakeComm* c, const char* what, bool (*ready)(const FakeComm*, int), int arg) {
  const double deadline = now_s() + timeout_s();
  unsigned spins = 0;
  while (!ready(c, arg)) {
    if ((++spins & 63u) == 0) {
      if (now_s() > deadline) die(c, what);
      timespec nap{0, 50000};
      nanosleep(&nap, nullptr);
    } else {
      sched_yield();
    }
  }
}

// generation-counting barrier over the shared header
void barrier(FakeComm* c) {
  const int gen = c->hdr->generation.load(std::memory_order_acquire);
  if (c->hdr->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->n_ranks) {
    c->hdr->arrived.store(0, std::memory_order_relaxed);
    c->hdr->generation.fetch_add(1, std::memory_order_acq_rel);
    return;
  }
  wait_until(c, "a rank did not reach the collective's barrier in time",
             [](const FakeComm* cc, int g) { return cc->hdr->generation.load(std::memory_order_acquire) != g; }, gen);
}

size_t elem_bytes(ncclDataType_t dt) {
  switch (dt) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

// The one data movement both collectives share: every rank's `bytes` bytes travel through its
// slot of the segment, chunk by chunk; ranks with `recv` copy every rank's chunk to
// recv + r * bytes + offset.
ncclResult_t exchange(FakeComm* c, const void* send, void* recv, size_t bytes, hipStream_t stream) {
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  const size_t chunks = bytes ? (bytes + c->slot_bytes - 1) / c->slot_bytes : 1;   // an empty call still synchronises the ranks
  for (size_t k = 0; k < chunks; ++k) {
    const size_t off = k * c->slot_bytes;
    const size_t n = bytes - off < c->slot_bytes ? bytes - off : c->slot_bytes;
    if (n && hipMemcpy(c->data + (size_t)c->rank * c->slot_bytes, (const char*)send + off, n, hipMemcpyDeviceToHost) != hipSuccess)
      return ncclUnhandledCudaError;
    barrier(c);                                          // every slot is filled
    if (recv && n)
      for (int r = 0; r < c->n_ranks; ++r)
        if (hipMemcpy((char*)recv + (size_t)r * bytes + off, c->data + (size_t)r * c->slot_bytes, n, hipMemcpyHostToDevice) != hipSuccess)
          return ncclUnhandledCudaError;
    barrier(c);                                          // every reader is done: slots may be refilled
  }
  if (c->rank == 0) c->hdr->collectives.fetch_add(1, std::memory_order_relaxed);
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  static std::atomic<unsigned> counter{0};
  memset(id, 0, sizeof(*id));
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(id->internal, sizeof(id->internal), "/smmfake-%d-%lld-%u", (int)getpid(),
           (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec, counter.fetch_add(1));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks <= 0 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  if (strncmp(id.internal, "/smmfake-", 9) != 0 || memchr(id.internal, 0, 64) == nullptr) return ncclInvalidArgument;
  FakeComm* c = new FakeComm();
  c->n_ranks = nranks;
  c->rank = rank;
  c->slot_bytes = slot_bytes_from_env();
  memcpy(c->name, id.internal, sizeof(c->name) - 1);   // NUL within the first 64 bytes was checked above
  c->map_bytes = sizeof(Header) + 4096 + (size_t)nranks * c->slot_bytes;
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
    if (fd >= 0) close(fd);
    delete c;
    return ncclSystemError;
  }
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    delete c;
    return ncclSystemError;
  }
  c->hdr = (Header*)p;
  c->data = (unsigned char*)p + 4096;
  static_assert(sizeof(Header) <= 4096, "header fits its page");
  c->hdr->magic.store(kMagic);
  c->hdr->attached.fetch_add(1, std::memory_order_acq_rel);
  // like the real call, this returns once every rank has joined
  wait_until(c, "not every rank called ncclCommInitRank in time",
             [](const FakeComm* cc, int n) { return cc->hdr->attached.load(std::memory_order_acquire) >= n; }, nranks);
  barrier(c);
  if (rank == 0) shm_unlink(c->name);   // the mappings live on; a crashed run leaves nothing behind in /dev/shm
  *comm = (ncclComm_t)c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  FakeComm* c = (FakeComm*)comm;
  if (!c) return ncclSuccess;
  c->hdr->detached.fetch_add(1);
  munmap((void*)c->hdr, c->map_bytes);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, int root,
                        ncclComm_t comm, hipStream_t stream) {
  FakeComm* c = (FakeComm*)comm;
  const size_t es = elem_bytes(datatype);
  if (!c || !es || root < 0 || root >= c->n_ranks) return ncclInvalidArgument;
  if (c->rank == root && !recvbuff && sendcount) return ncclInvalidArgument;
  return exchange(c, sendbuff, c->rank == root ? recvbuff : nullptr, sendcount * es, stream);
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype,
                           ncclComm_t comm, hipStream_t stream) {
  FakeComm* c = (FakeComm*)comm;
  const size_t es = elem_bytes(datatype);
  if (!c || !es || (!recvbuff && sendcount)) return ncclInvalidArgument;
  return exchange(c, sendbuff, recvbuff, sendcount * es, stream);
}

const char* ncclGetErrorString(ncclResult_t result) {
  switch (result) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "fake_rccl: a HIP call failed";
    case ncclSystemError: return "fake_rccl: shm_open / mmap failed";
    case ncclInvalidArgument: return "fake_rccl: invalid argument";
    default: return "fake_rccl: error";
  }
}

}  // extern "C"
