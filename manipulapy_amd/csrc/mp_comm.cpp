// RCCL communicator for the one-process-per-GPU multi-GPU path (include/manipula_hip.h, mp_comm_*).
// librccl.so is opened lazily so that single-GPU use and the CPU-only symbol tests never need it.
// The only collective on this path is the all-gather that reassembles the sharded torque history;
// trajectory batches are independent, so there is no exchange during compute.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/manipula_hip.h"

hipStream_t mp_ctx_compute_stream(mp_ctx* ctx);
int mp_ctx_device(mp_ctx* ctx);
int mp_set_error(int code, const char* msg);

namespace {
struct UniqueId { char internal[MP_UNIQUE_ID_BYTES]; };  // == ncclUniqueId (rccl.h:43)
typedef void* Comm;
struct Api {
  void* handle = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Api g_api;

int fail(const char* what, const char* detail) {
  char buf[400];
  std::snprintf(buf, sizeof buf, "%s: %s", what, detail ? detail : "?");
  return mp_set_error(MP_ERR_COMM, buf);
}

int load_api() {
  if (g_api.handle) return MP_OK;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) return fail("dlopen(librccl.so)", dlerror());
  Api a;
  a.handle = h;
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString)
    return fail("librccl.so", "missing nccl* symbol");
  g_api = a;
  return MP_OK;
}
int nccl_fail(const char* what, int rc) { return fail(what, g_api.GetErrorString ? g_api.GetErrorString(rc) : "?"); }
}  // namespace

struct mp_comm {
  mp_ctx* ctx = nullptr;
  Comm comm = nullptr;
  int nranks = 0, rank = 0;
};

extern "C" {

int mp_comm_unique_id(uint8_t id[MP_UNIQUE_ID_BYTES]) {
  if (!id) return mp_set_error(MP_ERR_INVALID, "mp_comm_unique_id: null output");
  if (int rc = load_api()) return rc;
  UniqueId u;
  std::memset(&u, 0, sizeof u);
  if (int rc = g_api.GetUniqueId(&u)) return nccl_fail("ncclGetUniqueId", rc);
  std::memcpy(id, u.internal, MP_UNIQUE_ID_BYTES);
  return MP_OK;
}

int mp_comm_create(mp_ctx* ctx, const uint8_t id[MP_UNIQUE_ID_BYTES], int nranks, int rank, mp_comm** out) {
  if (!ctx || !id || !out) return mp_set_error(MP_ERR_INVALID, "mp_comm_create: null argument");
  *out = nullptr;
  if (nranks < 1 || rank < 0 || rank >= nranks) return mp_set_error(MP_ERR_INVALID, "mp_comm_create: bad rank / nranks");
  if (int rc = load_api()) return rc;
  if (hipSetDevice(mp_ctx_device(ctx)) != hipSuccess) return mp_set_error(MP_ERR_HIP, "mp_comm_create: hipSetDevice failed");
  mp_comm* c = new (std::nothrow) mp_comm;
  if (!c) return mp_set_error(MP_ERR_INVALID, "mp_comm_create: out of host memory");
  c->ctx = ctx; c->nranks = nranks; c->rank = rank;
  UniqueId u;
  std::memcpy(u.internal, id, MP_UNIQUE_ID_BYTES);
  if (int rc = g_api.CommInitRank(&c->comm, nranks, u, rank)) { delete c; return nccl_fail("ncclCommInitRank", rc); }
  *out = c;
  return MP_OK;
}

int mp_comm_destroy(mp_comm* comm) {
  if (!comm) return MP_OK;
  if (comm->comm && g_api.CommDestroy) (void)g_api.CommDestroy(comm->comm);
  delete comm;
  return MP_OK;
}

int mp_comm_allgather(mp_comm* comm, const void* d_send, void* d_recv, size_t bytes_per_rank) {
  if (!comm || !d_send || !d_recv) return mp_set_error(MP_ERR_INVALID, "mp_comm_allgather: null argument");
  if (bytes_per_rank == 0) return MP_OK;
  if (hipSetDevice(mp_ctx_device(comm->ctx)) != hipSuccess) return mp_set_error(MP_ERR_HIP, "mp_comm_allgather: hipSetDevice failed");
  // ncclInt8 == 0 (rccl.h:459): the payload is opaque bytes
  if (int rc = g_api.AllGather(d_send, d_recv, bytes_per_rank, 0, comm->comm, mp_ctx_compute_stream(comm->ctx)))
    return nccl_fail("ncclAllGather", rc);
  return MP_OK;
}

}  // extern "C"
