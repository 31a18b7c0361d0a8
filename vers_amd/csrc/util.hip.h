// util.hip.h -- small host/device helpers shared by the translation units of libvers_hip.so.
#pragma once
#include "common.hpp"

namespace vers {

struct DeviceGuard {
  int prev = 0;
  explicit DeviceGuard(int dev) {
    (void)hipGetDevice(&prev);
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~DeviceGuard() { (void)hipSetDevice(prev); }
};

template <class T>
int32_t grow(T*& p, size_t& cap, size_t need) {
  if (need <= cap) return VERS_OK;
  if (p) VERS_HIP_TRY(hipFree(p));
  p = nullptr;
  cap = 0;
  VERS_HIP_TRY(hipMalloc((void**)&p, need * sizeof(T)));
  cap = need;
  return VERS_OK;
}

// rows [b][ld_in] (first d columns valid) -> blocks of qg rows, out[(g*ld_out + j)*qg + qi], zero padded
// (qg == 1: plain [b][ld_out]).  This is the query layout tile_chunk_compute reads through the scalar path.
static __global__ void stage_queries_kernel(const float* in, uint64_t ld_in, uint32_t d, float* out, uint32_t ld_out,
                                            uint32_t b, uint32_t qg) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n_groups = (b + qg - 1) / qg;
  if (i >= (uint64_t)n_groups * ld_out * qg) return;
  const uint32_t qi = (uint32_t)(i % qg);
  const uint32_t j = (uint32_t)((i / qg) % ld_out);
  const uint32_t q = (uint32_t)(i / ((uint64_t)qg * ld_out)) * qg + qi;
  out[i] = (j < d && q < b) ? in[(uint64_t)q * ld_in + j] : 0.0f;
}

inline int32_t launch_stage_queries(const float* in, uint64_t ld_in, uint32_t d, float* out, uint32_t ld_out, uint32_t b,
                                    uint32_t qg, hipStream_t st) {
  const uint64_t tot = (uint64_t)((b + qg - 1) / qg) * ld_out * qg;
  if (tot == 0) return VERS_OK;
  hipLaunchKernelGGL(stage_queries_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, in, ld_in, d, out, ld_out,
                     b, qg);
  VERS_HIP_TRY(hipGetLastError());
  return VERS_OK;
}

// ---- row-major <-> lane-transposed tiles (layout: scan.hip.h, blocked_index) -----------------------
// in: [n][ld_in] row-major (first d columns valid).  out: ceil(n/64) tiles of 64 x ld; rows >= n and
// columns >= d are zero.  One thread per output float4.
static __global__ void to_blocked_kernel(const float* in, uint64_t ld_in, uint32_t d, uint64_t n, float* out, uint32_t ld) {
  const uint32_t ld4 = ld / 4;
  const uint64_t n_pad = (n + 63) / 64 * 64;
  const uint64_t total = n_pad * ld4;
  for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t tile = s / (64ull * ld4);
    const uint32_t g = (uint32_t)(s % (64ull * ld4));
    const uint32_t j4 = g / 64, r = g % 64;
    const uint64_t row = tile * 64 + r;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (row < n) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j4 * 4 + u < d) v[u] = in[row * ld_in + j4 * 4 + u];
    }
    reinterpret_cast<float4*>(out)[s] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// rows [row0, row0+n) of a blocked matrix -> row-major out [n][ld_out] (first d columns)
static __global__ void from_blocked_kernel(const float* in, uint32_t ld, uint64_t row0, uint64_t n, uint32_t d, float* out,
                                           uint64_t ld_out) {
  const uint64_t total = n * d;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = i / d;
    const uint32_t j = (uint32_t)(i % d);
    const uint64_t row = row0 + r;
    out[r * ld_out + j] = in[(row >> 6) * 64ull * ld + ((uint64_t)(j >> 2) * 64 + (row & 63)) * 4 + (j & 3)];
  }
}

inline int32_t launch_to_blocked(const float* in, uint64_t ld_in, uint32_t d, uint64_t n, float* out, uint32_t ld,
                                 hipStream_t st) {
  if (n == 0) return VERS_OK;
  uint64_t blocks = ((n + 63) / 64 * 64 * (ld / 4) + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(to_blocked_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, ld_in, d, n, out, ld);
  VERS_HIP_TRY(hipGetLastError());
  return VERS_OK;
}

inline int32_t launch_from_blocked(const float* in, uint32_t ld, uint64_t row0, uint64_t n, uint32_t d, float* out,
                                   uint64_t ld_out, hipStream_t st) {
  if (n == 0) return VERS_OK;
  uint64_t blocks = (n * d + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(from_blocked_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, ld, row0, n, d, out, ld_out);
  VERS_HIP_TRY(hipGetLastError());
  return VERS_OK;
}

inline uint64_t blocked_floats(uint64_t n_rows, uint32_t ld) { return (n_rows + 63) / 64 * 64 * (uint64_t)ld; }

}  // namespace vers
