// Gradient exchange behind the C ABI: the flat fp32 gradient buffer is summed over the data-parallel ranks by RCCL's
// ncclAllReduce on the caller's stream (one process per GPU over xGMI) — the replacement for the CPU-side
// average_gradients of src/e2eflow/core/train.py:388-422 (concat + reduce_mean over the towers; the 1 / world factor is fused
// into the Adam kernel).  RCCL is resolved at run time (dlopen of the copy PyTorch already mapped, else librccl.so): the
// library loads and every other entry point works on a host without RCCL; these return UNFLOW_ERR_UNSUPPORTED there.
// The communicator is created from a 128-byte unique id that rank 0 obtains (unflow_comm_unique_id) and the host layer
// hands to the other ranks through whatever rendezvous it has (unflow_amd/core/data_parallel.py uses the torch.distributed store).
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "common.h"

namespace {

typedef struct { char internal[128]; } UniqueId;          // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                                        // ncclComm_t
constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0;              // ncclFloat32, ncclSum (rccl.h)

struct Rccl {
  void* handle = nullptr;
  int (*GetVersion)(int*) = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*CommCount)(Comm, int*) = nullptr;
  int (*CommUserRank)(Comm, int*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl q;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)
      if (!q.handle) q.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);      // the copy already in the process (torch's)
    for (const char* n : names)
      if (!q.handle) q.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!q.handle) return q;
#define SYM(field, name) q.field = reinterpret_cast<decltype(q.field)>(dlsym(q.handle, name))
    SYM(GetVersion, "ncclGetVersion");
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(CommCount, "ncclCommCount");
    SYM(CommUserRank, "ncclCommUserRank");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    q.ok = q.GetUniqueId && q.CommInitRank && q.CommDestroy && q.CommCount && q.CommUserRank && q.AllReduce;
    return q;
  }();
  return r;
}

int report(int rc, const char* what) {
  if (rc == 0) return UNFLOW_OK;
  Rccl& r = rccl();
  fprintf(stderr, "unflow: %s failed: %s\n", what, r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
  return UNFLOW_ERR_LAUNCH;
}

}  // namespace

UNFLOW_API int unflow_comm_available(void) {
  Rccl& r = rccl();
  if (!r.ok) return 0;
  int v = 0;
  if (r.GetVersion && r.GetVersion(&v) == 0 && v > 0) return v;
  return 1;
}

UNFLOW_API int unflow_comm_unique_id(void* id128) {
  if (!id128) return UNFLOW_ERR_NULL;
  Rccl& r = rccl();
  if (!r.ok) return UNFLOW_ERR_UNSUPPORTED;
  UniqueId id;
  const int rc = report(r.GetUniqueId(&id), "ncclGetUniqueId");
  if (rc == UNFLOW_OK) memcpy(id128, id.internal, sizeof(id.internal));
  return rc;
}

UNFLOW_API int unflow_comm_init(const void* id128, int nranks, int rank, void** comm) {
  if (!id128 || !comm) return UNFLOW_ERR_NULL;
  if (nranks < 1 || rank < 0 || rank >= nranks) return UNFLOW_ERR_SHAPE;
  Rccl& r = rccl();
  if (!r.ok) return UNFLOW_ERR_UNSUPPORTED;
  UniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  Comm c = nullptr;
  const int rc = report(r.CommInitRank(&c, nranks, id, rank), "ncclCommInitRank");
  *comm = rc == UNFLOW_OK ? c : nullptr;
  return rc;
}

UNFLOW_API int unflow_comm_info(void* comm, int* nranks, int* rank) {
  if (!comm) return UNFLOW_ERR_NULL;
  Rccl& r = rccl();
  if (!r.ok) return UNFLOW_ERR_UNSUPPORTED;
  int n = 0, k = 0;
  int rc = report(r.CommCount(comm, &n), "ncclCommCount");
  if (rc == UNFLOW_OK) rc = report(r.CommUserRank(comm, &k), "ncclCommUserRank");
  if (nranks) *nranks = n;
  if (rank) *rank = k;
  return rc;
}

// buf[0 .. n) <- sum over the ranks of `comm`, in place, enqueued on `stream` (returns at once; stream order is the only
// synchronisation).  Every rank must call it with the same n in the same order.
UNFLOW_API int unflow_allreduce_sum_f32(float* buf, long n, void* comm, unflow_stream_t stream) {
  if (!buf || !comm) return UNFLOW_ERR_NULL;
  if (n < 0) return UNFLOW_ERR_SHAPE;
  if (n == 0) return UNFLOW_OK;
  Rccl& r = rccl();
  if (!r.ok) return UNFLOW_ERR_UNSUPPORTED;
  return report(r.AllReduce(buf, buf, (size_t)n, NCCL_FLOAT32, NCCL_SUM, comm, as_stream(stream)), "ncclAllReduce");
}

UNFLOW_API int unflow_comm_destroy(void* comm) {
  if (!comm) return UNFLOW_OK;
  Rccl& r = rccl();
  if (!r.ok) return UNFLOW_ERR_UNSUPPORTED;
  return report(r.CommDestroy(comm), "ncclCommDestroy");
}
