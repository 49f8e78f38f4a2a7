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
int mp_ctx_flush_parked(mp_ctx* ctx);  // runs the float64 passes parked behind float32 launches; its failure fails the collective
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
  int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
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
  a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(h, "ncclSend"));
  a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(h, "ncclRecv"));
  a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
  a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.GetErrorString || !a.Send || !a.Recv ||
      !a.GroupStart || !a.GroupEnd)
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
  hipStream_t stream = nullptr;  // the exchanges of mp_comm_exchange_chunk run here, beside the compute stream
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
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    (void)g_api.CommDestroy(c->comm);
    delete c;
    return mp_set_error(MP_ERR_HIP, "mp_comm_create: hipStreamCreate failed");
  }
  *out = c;
  return MP_OK;
}

int mp_comm_destroy(mp_comm* comm) {
  if (!comm) return MP_OK;
  if (comm->stream) { (void)hipStreamSynchronize(comm->stream); (void)hipStreamDestroy(comm->stream); }
  if (comm->comm && g_api.CommDestroy) (void)g_api.CommDestroy(comm->comm);
  delete comm;
  return MP_OK;
}

int mp_comm_allgather(mp_comm* comm, const void* d_send, void* d_recv, size_t bytes_per_rank) {
  if (!comm || !d_send || !d_recv) return mp_set_error(MP_ERR_INVALID, "mp_comm_allgather: null argument");
  if (bytes_per_rank == 0) return MP_OK;
  if (hipSetDevice(mp_ctx_device(comm->ctx)) != hipSuccess) return mp_set_error(MP_ERR_HIP, "mp_comm_allgather: hipSetDevice failed");
  if (int rc = mp_ctx_flush_parked(comm->ctx)) return rc;
  // ncclInt8 == 0 (rccl.h:459): the payload is opaque bytes
  if (int rc = g_api.AllGather(d_send, d_recv, bytes_per_rank, 0, comm->comm, mp_ctx_compute_stream(comm->ctx)))
    return nccl_fail("ncclAllGather", rc);
  return MP_OK;
}

// Uneven shards (B % world != 0): rank r contributes bytes_of_rank[r] bytes, d_recv holds the shards back to back in rank order.
// ncclAllGather wants equal counts, so this is the grouped point-to-point form (one ncclSend + ncclRecv per peer in one group:
// xGMI is point to point, all links carry a share at once) on the compute stream, plus a device copy of this rank's own shard.
int mp_comm_allgatherv(mp_comm* comm, const void* d_send, void* d_recv, const size_t* bytes_of_rank) {
  if (!comm || !d_recv || !bytes_of_rank) return mp_set_error(MP_ERR_INVALID, "mp_comm_allgatherv: null argument");
  if (hipSetDevice(mp_ctx_device(comm->ctx)) != hipSuccess) return mp_set_error(MP_ERR_HIP, "mp_comm_allgatherv: hipSetDevice failed");
  if (int rc = mp_ctx_flush_parked(comm->ctx)) return rc;
  hipStream_t s = mp_ctx_compute_stream(comm->ctx);
  char* base = static_cast<char*>(d_recv);
  size_t mine_off = 0;
  for (int p = 0; p < comm->rank; ++p) mine_off += bytes_of_rank[p];
  const size_t mine = bytes_of_rank[comm->rank];
  if (mine && !d_send) return mp_set_error(MP_ERR_INVALID, "mp_comm_allgatherv: null send buffer");
  if (mine && d_send != base + mine_off &&
      hipMemcpyAsync(base + mine_off, d_send, mine, hipMemcpyDeviceToDevice, s) != hipSuccess)
    return mp_set_error(MP_ERR_HIP, "mp_comm_allgatherv: device copy of the local shard failed");
  if (comm->nranks == 1) return MP_OK;
  if (int rc = g_api.GroupStart()) return nccl_fail("ncclGroupStart", rc);
  int err = 0;
  size_t off = 0;
  for (int p = 0; p < comm->nranks && !err; off += bytes_of_rank[p], ++p) {
    if (p == comm->rank) continue;
    if (mine) err = g_api.Send(d_send, mine, 0, p, comm->comm, s);
    if (!err && bytes_of_rank[p]) err = g_api.Recv(base + off, bytes_of_rank[p], 0, p, comm->comm, s);
  }
  const int end = g_api.GroupEnd();
  if (err) return nccl_fail("ncclSend / ncclRecv", err);
  if (end) return nccl_fail("ncclGroupEnd", end);
  return MP_OK;
}

// order stream `after` behind everything queued so far on stream `before`
static int chain(hipStream_t before, hipStream_t after, const char* fn) {
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return mp_set_error(MP_ERR_HIP, fn);
  hipError_t e = hipEventRecord(ev, before);
  if (e == hipSuccess) e = hipStreamWaitEvent(after, ev, 0);
  (void)hipEventDestroy(ev);  // released by the runtime once the recorded work has completed
  return e == hipSuccess ? MP_OK : mp_set_error(MP_ERR_HIP, fn);
}

int mp_comm_exchange_chunk(mp_comm* comm, void* d_all, size_t bytes_per_rank, size_t offset, size_t nbytes) {
  if (!comm || !d_all) return mp_set_error(MP_ERR_INVALID, "mp_comm_exchange_chunk: null argument");
  if (offset + nbytes > bytes_per_rank) return mp_set_error(MP_ERR_INVALID, "mp_comm_exchange_chunk: chunk exceeds the per-rank slot");
  if (nbytes == 0) return MP_OK;
  if (hipSetDevice(mp_ctx_device(comm->ctx)) != hipSuccess) return mp_set_error(MP_ERR_HIP, "mp_comm_exchange_chunk: hipSetDevice failed");
  // the chunk was produced by kernels already queued on the compute stream: the exchange starts when they are done,
  // on its own stream, so the kernels of the NEXT chunk run beside it
  if (int rc = mp_ctx_flush_parked(comm->ctx)) return rc;
  if (int rc = chain(mp_ctx_compute_stream(comm->ctx), comm->stream, "mp_comm_exchange_chunk: event chain failed")) return rc;
  if (comm->nranks == 1) return MP_OK;
  char* base = static_cast<char*>(d_all);
  if (int rc = g_api.GroupStart()) return nccl_fail("ncclGroupStart", rc);
  int err = 0;
  for (int p = 0; p < comm->nranks && !err; ++p) {
    if (p == comm->rank) continue;
    // xGMI is point to point: one send and one receive per peer, all seven links busy at once (no ring)
    err = g_api.Send(base + (size_t)comm->rank * bytes_per_rank + offset, nbytes, 0, p, comm->comm, comm->stream);
    if (!err) err = g_api.Recv(base + (size_t)p * bytes_per_rank + offset, nbytes, 0, p, comm->comm, comm->stream);
  }
  const int end = g_api.GroupEnd();
  if (err) return nccl_fail("ncclSend / ncclRecv", err);
  if (end) return nccl_fail("ncclGroupEnd", end);
  return MP_OK;
}

// The overlapped exchange for uneven shards: rank r's slot starts slot_offset[r] bytes into d_all and this chunk covers
// [chunk_offset[r], chunk_offset[r] + chunk_bytes[r]) of it (every rank passes the same three arrays, nranks entries each).
int mp_comm_exchange_chunk_v(mp_comm* comm, void* d_all, const size_t* slot_offset, const size_t* chunk_offset, const size_t* chunk_bytes) {
  if (!comm || !d_all || !slot_offset || !chunk_offset || !chunk_bytes) return mp_set_error(MP_ERR_INVALID, "mp_comm_exchange_chunk_v: null argument");
  if (hipSetDevice(mp_ctx_device(comm->ctx)) != hipSuccess) return mp_set_error(MP_ERR_HIP, "mp_comm_exchange_chunk_v: hipSetDevice failed");
  if (int rc = mp_ctx_flush_parked(comm->ctx)) return rc;
  if (int rc = chain(mp_ctx_compute_stream(comm->ctx), comm->stream, "mp_comm_exchange_chunk_v: event chain failed")) return rc;
  if (comm->nranks == 1) return MP_OK;
  char* base = static_cast<char*>(d_all);
  const int me = comm->rank;
  if (int rc = g_api.GroupStart()) return nccl_fail("ncclGroupStart", rc);
  int err = 0;
  for (int p = 0; p < comm->nranks && !err; ++p) {
    if (p == me) continue;
    if (chunk_bytes[me]) err = g_api.Send(base + slot_offset[me] + chunk_offset[me], chunk_bytes[me], 0, p, comm->comm, comm->stream);
    if (!err && chunk_bytes[p]) err = g_api.Recv(base + slot_offset[p] + chunk_offset[p], chunk_bytes[p], 0, p, comm->comm, comm->stream);
  }
  const int end = g_api.GroupEnd();
  if (err) return nccl_fail("ncclSend / ncclRecv", err);
  if (end) return nccl_fail("ncclGroupEnd", end);
  return MP_OK;
}

int mp_comm_join(mp_comm* comm) {
  if (!comm) return mp_set_error(MP_ERR_INVALID, "mp_comm_join: null communicator");
  if (hipSetDevice(mp_ctx_device(comm->ctx)) != hipSuccess) return mp_set_error(MP_ERR_HIP, "mp_comm_join: hipSetDevice failed");
  return chain(comm->stream, mp_ctx_compute_stream(comm->ctx), "mp_comm_join: event chain failed");
}

}  // extern "C"
