// vers_comm_rccl.hip -- libvers_rccl.so: the multi-GPU exchanges of include/vers_hip.h over an RCCL communicator
// (include/vers_comm_rccl.h).  Host code only: every function queues RCCL calls; the kernels are RCCL's.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

#include "../../../include/vers_comm_rccl.h"

static_assert(VERS_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id the host passes around is an ncclUniqueId");

namespace {
thread_local std::string g_err;
int32_t fail(int32_t status, const std::string& msg) {
  g_err = msg;
  return status;
}
#define RCCL_TRY(expr)                                                                                    \
  do {                                                                                                    \
    ncclResult_t _r = (expr);                                                                             \
    if (_r != ncclSuccess) return fail(VERS_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(_r)); \
  } while (0)
#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) return fail(VERS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
struct DeviceGuard {
  int prev = 0;
  explicit DeviceGuard(int dev) {
    (void)hipGetDevice(&prev);
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~DeviceGuard() { (void)hipSetDevice(prev); }
};
}  // namespace

struct vers_rccl {
  ncclComm_t comm = nullptr;
  bool owned = false;
  int device = 0;
  uint32_t rank = 0, world = 1;
  hipStream_t stream = nullptr;  // the synchronous callbacks' own stream
};

namespace {

// ---- the search's exchange: stream-ordered, nothing waits ------------------------------------------------------------
int32_t gather_async(void* ctx, const void* send_dev, void* recv_dev, uint64_t bytes, void* stream) {
  vers_rccl* c = (vers_rccl*)ctx;
  // (the caller -- vers_ivf_search_sharded_dev -- has the device current)
  RCCL_TRY(ncclAllGather(send_dev, recv_dev, (size_t)bytes, ncclUint8, c->comm, (hipStream_t)stream));
  return VERS_OK;
}

// ---- the build's exchanges: queue on the handle's stream, wait ----------------------------------------------------------
int32_t cb_all_gather(void* ctx, const void* send_dev, void* recv_dev, uint64_t bytes) {
  vers_rccl* c = (vers_rccl*)ctx;
  DeviceGuard g(c->device);
  RCCL_TRY(ncclAllGather(send_dev, recv_dev, (size_t)bytes, ncclUint8, c->comm, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return VERS_OK;
}
int32_t cb_send(void* ctx, const void* buf_dev, uint64_t bytes, uint32_t peer) {
  vers_rccl* c = (vers_rccl*)ctx;
  DeviceGuard g(c->device);
  RCCL_TRY(ncclSend(buf_dev, (size_t)bytes, ncclUint8, (int)peer, c->comm, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return VERS_OK;
}
int32_t cb_recv(void* ctx, void* buf_dev, uint64_t bytes, uint32_t peer) {
  vers_rccl* c = (vers_rccl*)ctx;
  DeviceGuard g(c->device);
  RCCL_TRY(ncclRecv(buf_dev, (size_t)bytes, ncclUint8, (int)peer, c->comm, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return VERS_OK;
}
int32_t cb_broadcast(void* ctx, void* buf_dev, uint64_t bytes, uint32_t root) {
  vers_rccl* c = (vers_rccl*)ctx;
  DeviceGuard g(c->device);
  RCCL_TRY(ncclBroadcast(buf_dev, buf_dev, (size_t)bytes, ncclUint8, (int)root, c->comm, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return VERS_OK;
}
// rows to the owners of their lists: ONE grouped exchange -- every send and receive of the rank is posted inside a single
// ncclGroupStart / ncclGroupEnd, so no ordering between peers can deadlock (the rank's own share is a device copy)
int32_t cb_all_to_all_v(void* ctx, const void* send_dev, const uint64_t* send_bytes, const uint64_t* send_off, void* recv_dev,
                        const uint64_t* recv_bytes, const uint64_t* recv_off) {
  vers_rccl* c = (vers_rccl*)ctx;
  DeviceGuard g(c->device);
  if (send_bytes[c->rank] != recv_bytes[c->rank]) return fail(VERS_ERR_INVALID, "all_to_all_v: a rank's share for itself differs between its send and receive plans");
  if (send_bytes[c->rank])
    HIP_TRY(hipMemcpyAsync((char*)recv_dev + recv_off[c->rank], (const char*)send_dev + send_off[c->rank], (size_t)send_bytes[c->rank],
                           hipMemcpyDeviceToDevice, c->stream));
  RCCL_TRY(ncclGroupStart());
  for (uint32_t p = 0; p < c->world; ++p) {
    if (p == c->rank) continue;
    if (send_bytes[p]) RCCL_TRY(ncclSend((const char*)send_dev + send_off[p], (size_t)send_bytes[p], ncclUint8, (int)p, c->comm, c->stream));
    if (recv_bytes[p]) RCCL_TRY(ncclRecv((char*)recv_dev + recv_off[p], (size_t)recv_bytes[p], ncclUint8, (int)p, c->comm, c->stream));
  }
  RCCL_TRY(ncclGroupEnd());
  HIP_TRY(hipStreamSynchronize(c->stream));
  return VERS_OK;
}

int32_t finish_create(vers_rccl* c, vers_rccl_t** out) {
  int n = 0, r = 0;
  RCCL_TRY(ncclCommCount(c->comm, &n));
  RCCL_TRY(ncclCommUserRank(c->comm, &r));
  c->world = (uint32_t)n;
  c->rank = (uint32_t)r;
  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  *out = c;
  return VERS_OK;
}

}  // namespace

extern "C" {

const char* vers_rccl_last_error(void) { return g_err.c_str(); }

int32_t vers_rccl_unique_id(void* out_id128) {
  if (!out_id128) return fail(VERS_ERR_INVALID, "vers_rccl_unique_id: null buffer");
  ncclUniqueId id;
  RCCL_TRY(ncclGetUniqueId(&id));
  std::memcpy(out_id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return VERS_OK;
}

int32_t vers_rccl_create(const void* id128, uint32_t rank, uint32_t world, int32_t device, vers_rccl_t** out) {
  if (!id128 || !out || world == 0 || world > 255 || rank >= world) return fail(VERS_ERR_INVALID, "vers_rccl_create: bad arguments");
  int cnt = 0;
  HIP_TRY(hipGetDeviceCount(&cnt));
  if (device < 0 || device >= cnt) return fail(VERS_ERR_INVALID, "vers_rccl_create: no such device");
  DeviceGuard g(device);
  vers_rccl* c = new (std::nothrow) vers_rccl();
  if (!c) return fail(VERS_ERR_INVALID, "out of host memory");
  c->device = device;
  c->owned = true;
  ncclUniqueId id;
  std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t r = ncclCommInitRank(&c->comm, (int)world, id, (int)rank);
  if (r != ncclSuccess) {
    delete c;
    return fail(VERS_ERR_COMM, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  }
  const int32_t rc = finish_create(c, out);
  if (rc) (void)vers_rccl_destroy(c);
  return rc;
}

int32_t vers_rccl_adopt(void* nccl_comm, int32_t device, vers_rccl_t** out) {
  if (!nccl_comm || !out) return fail(VERS_ERR_INVALID, "vers_rccl_adopt: bad arguments");
  DeviceGuard g(device);
  vers_rccl* c = new (std::nothrow) vers_rccl();
  if (!c) return fail(VERS_ERR_INVALID, "out of host memory");
  c->device = device;
  c->owned = false;
  c->comm = (ncclComm_t)nccl_comm;
  const int32_t rc = finish_create(c, out);
  if (rc) (void)vers_rccl_destroy(c);
  return rc;
}

int32_t vers_rccl_destroy(vers_rccl_t* c) {
  if (!c) return VERS_OK;
  DeviceGuard g(c->device);
  if (c->stream) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamDestroy(c->stream);
  }
  if (c->owned && c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
  return VERS_OK;
}

int32_t vers_rccl_gather(vers_rccl_t* c, vers_gather_t* out) {
  if (!c || !out) return fail(VERS_ERR_INVALID, "vers_rccl_gather: bad arguments");
  out->ctx = c;
  out->rank = c->rank;
  out->world = c->world;
  out->all_gather_async = gather_async;
  return VERS_OK;
}

int32_t vers_rccl_comm(vers_rccl_t* c, vers_comm_t* out) {
  if (!c || !out) return fail(VERS_ERR_INVALID, "vers_rccl_comm: bad arguments");
  out->ctx = c;
  out->rank = c->rank;
  out->world = c->world;
  out->all_gather = cb_all_gather;
  out->send = cb_send;
  out->recv = cb_recv;
  out->broadcast = cb_broadcast;
  out->all_to_all_v = cb_all_to_all_v;
  return VERS_OK;
}

}  // extern "C"
