// util.cuh -- small host/device helpers shared by the translation units of libvers_hip.so.
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

}  // namespace vers
