// vers_comm_rccl.hip -- libvers_rccl.so: the multi-GPU exchanges of include/vers_hip.h over an RCCL communicator
// (include/vers_comm_rccl.h).  Host code only: every function queues RCCL calls; the kernels are RCCL's.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

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
// makes `dev` current for a scope; ok == false: it could not (the caller returns VERS_ERR_HIP instead of working on whatever
// device happened to be current); the previous device is restored only if it was read successfully
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    if ((err = hipGetDevice(&prev)) != hipSuccess) { prev = -1; ok = false; return; }
    if (prev != dev && (err = hipSetDevice(dev)) != hipSuccess) ok = false;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define GUARD_TRY(g)                                                                                                        \
  do {                                                                                                                      \
    if (!(g).ok) return fail(VERS_ERR_HIP, std::string("hipSetDevice / hipGetDevice: ") + hipGetErrorString((g).err));      \
  } while (0)
#define ALIVE(c, what)                                                                                                                  \
  do {                                                                                                                                  \
    if ((c)->dead) return fail(VERS_ERR_COMM, std::string(what) + ": the communicator is dead (an earlier exchange timed out or failed)"); \
  } while (0)
}  // namespace

struct vers_rccl {
  ncclComm_t comm = nullptr;
  bool owned = false;
  bool dead = false;  // a synchronous exchange timed out or failed asynchronously: every later call fails at once
  int device = 0;
  uint32_t rank = 0, world = 1;
  hipStream_t stream = nullptr;  // the synchronous callbacks' own stream
};

namespace {

// ---- the search's exchange: stream-ordered, nothing waits ------------------------------------------------------------
int32_t gather_async(void* ctx, const void* send_dev, void* recv_dev, uint64_t bytes, void* stream) {
  vers_rccl* c = (vers_rccl*)ctx;
  ALIVE(c, "all_gather_async");
  // (the caller -- vers_ivf_search_sharded_dev -- has the device current)
  RCCL_TRY(ncclAllGather(send_dev, recv_dev, (size_t)bytes, ncclUint8, c->comm, (hipStream_t)stream));
  return VERS_OK;
}

// ---- the build's exchanges: queue on the handle's stream, wait ----------------------------------------------------------
// The wait is BOUNDED: a peer that died (or returned early from the build) never posts its half of the exchange, and a plain
// hipStreamSynchronize would keep this rank -- and with it the 6 others -- inside a spinning RCCL kernel for ever.  The stream
// is polled (hipStreamQuery, 50 us naps after the first millisecond) together with the communicator's asynchronous error state;
// past the deadline (VERS_RCCL_TIMEOUT_S, default 600 s: a k = 65536 pass of cfg5 keeps a peer busy for seconds, not minutes)
// the callback returns VERS_ERR_COMM, the handle is marked dead and the build returns non-zero on this rank: the host then
// aborts the communicator (vers_rccl_abort) and exits -- never a re-exec (a process that has initialised the GPU must not).
double timeout_seconds() {
  static const double t = [] { const char* e = getenv("VERS_RCCL_TIMEOUT_S"); const double v = e ? atof(e) : 600.0; return v > 0 ? v : 600.0; }();
  return t;
}
int32_t wait_stream(vers_rccl* c, const char* what) {
  const auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spins = 0;; ++spins) {
    const hipError_t q = hipStreamQuery(c->stream);
    if (q == hipSuccess) return VERS_OK;
    if (q != hipErrorNotReady) { c->dead = true; return fail(VERS_ERR_HIP, std::string(what) + ": hipStreamQuery: " + hipGetErrorString(q)); }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if ((spins & 63u) == 63u) {
      ncclResult_t async = ncclSuccess;
      if (ncclCommGetAsyncError(c->comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
        c->dead = true;
        return fail(VERS_ERR_COMM, std::string(what) + ": the communicator reports an asynchronous error: " + ncclGetErrorString(async));
      }
    }
    if (el > timeout_seconds()) {
      c->dead = true;
      return fail(VERS_ERR_COMM, std::string(what) + ": no completion after " + std::to_string((int)el) + " s (VERS_RCCL_TIMEOUT_S): a peer is gone or left the build; abort the communicator (vers_rccl_abort) and exit");
    }
    if (el > 1e-3) std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}
int32_t cb_all_gather(void* ctx, const void* send_dev, void* recv_dev, uint64_t bytes) {
  vers_rccl* c = (vers_rccl*)ctx;
  ALIVE(c, "all_gather");
  DeviceGuard g(c->device);
  GUARD_TRY(g);
  RCCL_TRY(ncclAllGather(send_dev, recv_dev, (size_t)bytes, ncclUint8, c->comm, c->stream));
  return wait_stream(c, "all_gather");
}
int32_t cb_send(void* ctx, const void* buf_dev, uint64_t bytes, uint32_t peer) {
  vers_rccl* c = (vers_rccl*)ctx;
  ALIVE(c, "send");
  DeviceGuard g(c->device);
  GUARD_TRY(g);
  RCCL_TRY(ncclSend(buf_dev, (size_t)bytes, ncclUint8, (int)peer, c->comm, c->stream));
  return wait_stream(c, "send");
}
int32_t cb_recv(void* ctx, void* buf_dev, uint64_t bytes, uint32_t peer) {
  vers_rccl* c = (vers_rccl*)ctx;
  ALIVE(c, "recv");
  DeviceGuard g(c->device);
  GUARD_TRY(g);
  RCCL_TRY(ncclRecv(buf_dev, (size_t)bytes, ncclUint8, (int)peer, c->comm, c->stream));
  return wait_stream(c, "recv");
}
int32_t cb_broadcast(void* ctx, void* buf_dev, uint64_t bytes, uint32_t root) {
  vers_rccl* c = (vers_rccl*)ctx;
  ALIVE(c, "broadcast");
  DeviceGuard g(c->device);
  GUARD_TRY(g);
  RCCL_TRY(ncclBroadcast(buf_dev, buf_dev, (size_t)bytes, ncclUint8, (int)root, c->comm, c->stream));
  return wait_stream(c, "broadcast");
}
// rows to the owners of their lists: ONE grouped exchange -- every send and receive of the rank is posted inside a single
// ncclGroupStart / ncclGroupEnd, so no ordering between peers can deadlock (the rank's own share is a device copy)
int32_t cb_all_to_all_v(void* ctx, const void* send_dev, const uint64_t* send_bytes, const uint64_t* send_off, void* recv_dev,
                        const uint64_t* recv_bytes, const uint64_t* recv_off) {
  vers_rccl* c = (vers_rccl*)ctx;
  ALIVE(c, "all_to_all_v");
  DeviceGuard g(c->device);
  GUARD_TRY(g);
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
  return wait_stream(c, "all_to_all_v");
}

// The adapter is compiled against ONE rccl.h and runs on whatever librccl the process mapped first (under PyTorch: torch's
// bundled copy, not /opt/rocm's).  The entry points used here -- ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclSend /
// ncclRecv, ncclBroadcast, group start / end, ncclCommCount / UserRank / CuDevice / Abort / Destroy / GetAsyncError -- have
// kept their signatures through NCCL 2.x; a different MAJOR version is refused, a different minor is reported once.
int32_t check_runtime_version() {
  int v = 0;
  RCCL_TRY(ncclGetVersion(&v));
  if (v / 10000 != NCCL_VERSION_CODE / 10000)
    return fail(VERS_ERR_COMM, "libvers_rccl.so was built against RCCL " + std::to_string(NCCL_VERSION_CODE) + " but the process loaded RCCL " + std::to_string(v) + ": incompatible major version");
  static bool said = false;
  if (v != NCCL_VERSION_CODE && !said) {
    said = true;
    Dl_info info;
    const char* path = dladdr((void*)&ncclGetVersion, &info) && info.dli_fname ? info.dli_fname : "?";
    fprintf(stderr, "[vers rccl] built against RCCL %d, running on RCCL %d (%s): same major version, 2.x entry points only\n", NCCL_VERSION_CODE, v, path);
  }
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
  if (int32_t rc = check_runtime_version()) return rc;
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
  GUARD_TRY(g);
  if (int32_t rc = check_runtime_version()) return rc;
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
  int cnt = 0;
  HIP_TRY(hipGetDeviceCount(&cnt));
  if (device < 0 || device >= cnt) return fail(VERS_ERR_INVALID, "vers_rccl_adopt: no such device");
  {  // the communicator was made on ONE device: the handle's stream and the callers' buffers must live there
    int comm_dev = -1;
    RCCL_TRY(ncclCommCuDevice((ncclComm_t)nccl_comm, &comm_dev));
    if (comm_dev != device)
      return fail(VERS_ERR_INVALID, "vers_rccl_adopt: the communicator lives on device " + std::to_string(comm_dev) + ", not on device " + std::to_string(device));
  }
  DeviceGuard g(device);
  GUARD_TRY(g);
  if (int32_t rc = check_runtime_version()) return rc;
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
  // A DEAD communicator (a bounded wait expired, an exchange failed) may still have an RCCL kernel on the stream spinning for the
  // peer that is gone.  hipStreamSynchronize AND hipStreamDestroy both drain the stream's queue on ROCm, i.e. either would wait for
  // that kernel forever -- the hang the bounded wait exists to avoid.  So the communicator is aborted FIRST (owned or adopted: as
  // vers_rccl_abort does; an adopted communicator that died must not be used by its maker again either), which ends its kernels;
  // only then is the stream touched.
  if (c->dead && c->comm) {
    (void)ncclCommAbort(c->comm);
    c->comm = nullptr;
  }
  if (c->stream) {
    if (!c->dead) (void)hipStreamSynchronize(c->stream);
    (void)hipStreamDestroy(c->stream);
  }
  if (c->owned && c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
  return VERS_OK;
}

int32_t vers_rccl_abort(vers_rccl_t* c) {
  if (!c) return fail(VERS_ERR_INVALID, "vers_rccl_abort: null handle");
  DeviceGuard g(c->device);
  c->dead = true;
  if (c->comm) {
    const ncclResult_t r = ncclCommAbort(c->comm);  // (frees the communicator: owned or adopted, it must not be used again)
    c->comm = nullptr;
    if (r != ncclSuccess) return fail(VERS_ERR_COMM, std::string("ncclCommAbort: ") + ncclGetErrorString(r));
  }
  return VERS_OK;
}

int32_t vers_rccl_versions(int32_t* out_build_code, int32_t* out_runtime_code, char* out_path, uint64_t path_cap) {
  if (out_build_code) *out_build_code = NCCL_VERSION_CODE;
  int v = 0;
  RCCL_TRY(ncclGetVersion(&v));
  if (out_runtime_code) *out_runtime_code = v;
  if (out_path && path_cap) {
    out_path[0] = 0;
    Dl_info info;
    if (dladdr((void*)&ncclGetVersion, &info) && info.dli_fname) {
      std::strncpy(out_path, info.dli_fname, (size_t)path_cap - 1);
      out_path[path_cap - 1] = 0;
    }
  }
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
