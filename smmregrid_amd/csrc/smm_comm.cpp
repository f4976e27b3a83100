// RCCL (xGMI) exchange of output shards behind the C ABI: smm_comm_* in
// include/smmregrid_amd.h.  The reference has no distributed layer at all
// (SURVEY section 5); this is the C1 piece of SURVEY section 2: a gather /
// all-gather of the Y shards of a batch-sharded regrid.
//
// librccl is bound at run time (dlsym on the process first, so a copy already
// loaded by e.g. torch is reused; else dlopen), which keeps single-GPU users
// free of the dependency.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/smmregrid_amd.h"

namespace smm {
int fail_msg(int code, const std::string& msg);  // smm_device.hip
}

namespace {

struct Rccl {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGather) Gather = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
  std::string err;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(RTLD_DEFAULT, name);
      if (!p && h) p = dlsym(h, name);
      return p;
    };
    if (!dlsym(RTLD_DEFAULT, "ncclGetUniqueId")) {
      // SMM_RCCL_LIB names the library to bind instead of the default search list
      const char* forced = getenv("SMM_RCCL_LIB");
      std::string why;
      auto try_open = [&](const char* path) {
        (void)dlerror();
        h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
          const char* e = dlerror();   // one call: dlerror() clears the message it returns
          why = e ? e : "not found";
        }
      };
      if (forced && *forced) {
        try_open(forced);
      } else {
        for (const char* path : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
          try_open(path);
          if (h) break;
        }
      }
      if (!h) {
        r.err = "cannot load librccl: " + why;
        return;
      }
    }
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.Gather = (decltype(r.Gather))sym("ncclGather");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Gather && r.AllGather &&
           r.GetErrorString;
    if (!r.ok) r.err = "librccl lacks a required symbol";
  });
  return r;
}

int check(ncclResult_t e, const char* what) {
  if (e == ncclSuccess) return SMM_OK;
  return smm::fail_msg(SMM_ERR_HIP, std::string(what) + ": " + rccl().GetErrorString(e));
}

bool dtype_of(int dtype, ncclDataType_t* out) {
  if (dtype == SMM_F32) *out = ncclFloat32;
  else if (dtype == SMM_F64) *out = ncclFloat64;
  else return false;
  return true;
}

}  // namespace

struct smm_comm {
  ncclComm_t comm = nullptr;
  int n_ranks = 0, rank = 0;
};

extern "C" {

int smm_comm_unique_id(void* id_out) {
  if (!id_out) return smm::fail_msg(SMM_ERR_INVALID, "null id buffer");
  Rccl& r = rccl();
  if (!r.ok) return smm::fail_msg(SMM_ERR_UNSUPPORTED, r.err);
  ncclUniqueId id;
  int rc = check(r.GetUniqueId(&id), "ncclGetUniqueId");
  if (rc) return rc;
  static_assert(sizeof(id) == SMM_COMM_ID_BYTES, "unique id size");
  memcpy(id_out, &id, sizeof(id));
  return SMM_OK;
}

int smm_comm_create(const void* id, int n_ranks, int rank, smm_comm_t* out) {
  if (!out) return smm::fail_msg(SMM_ERR_INVALID, "null out handle");
  *out = nullptr;
  if (!id || n_ranks <= 0 || rank < 0 || rank >= n_ranks)
    return smm::fail_msg(SMM_ERR_INVALID, "bad rank / size / id");
  Rccl& r = rccl();
  if (!r.ok) return smm::fail_msg(SMM_ERR_UNSUPPORTED, r.err);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  smm_comm* c = new (std::nothrow) smm_comm();
  if (!c) return smm::fail_msg(SMM_ERR_ALLOC, "out of host memory");
  c->n_ranks = n_ranks;
  c->rank = rank;
  int rc = check(r.CommInitRank(&c->comm, n_ranks, uid, rank), "ncclCommInitRank");
  if (rc) {
    delete c;
    return rc;
  }
  *out = c;
  return SMM_OK;
}

int smm_comm_destroy(smm_comm_t c) {
  if (!c) return SMM_OK;
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  delete c;
  return SMM_OK;
}

int smm_comm_gather(smm_comm_t c, const void* send_dev, void* recv_dev, int64_t count, int dtype,
                    int root, void* stream) {
  if (!c) return smm::fail_msg(SMM_ERR_INVALID, "null communicator");
  ncclDataType_t dt;
  if (!dtype_of(dtype, &dt)) return smm::fail_msg(SMM_ERR_UNSUPPORTED, "dtype must be SMM_F32 or SMM_F64");
  if (count < 0 || root < 0 || root >= c->n_ranks) return smm::fail_msg(SMM_ERR_INVALID, "bad count / root");
  if (c->rank == root && !recv_dev && count > 0) return smm::fail_msg(SMM_ERR_INVALID, "root needs a receive buffer");
  return check(rccl().Gather(send_dev, recv_dev, (size_t)count, dt, root, c->comm, (hipStream_t)stream),
               "ncclGather");
}

int smm_comm_allgather(smm_comm_t c, const void* send_dev, void* recv_dev, int64_t count, int dtype,
                       void* stream) {
  if (!c) return smm::fail_msg(SMM_ERR_INVALID, "null communicator");
  ncclDataType_t dt;
  if (!dtype_of(dtype, &dt)) return smm::fail_msg(SMM_ERR_UNSUPPORTED, "dtype must be SMM_F32 or SMM_F64");
  if (count < 0) return smm::fail_msg(SMM_ERR_INVALID, "negative count");
  return check(rccl().AllGather(send_dev, recv_dev, (size_t)count, dt, c->comm, (hipStream_t)stream),
               "ncclAllGather");
}

}  // extern "C"
