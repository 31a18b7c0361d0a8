// gen.hip -- synthetic corpus generator for bench.py, written straight into HBM (a 10M x 768 corpus
// cannot be staged through the host).  Restates tests/datagen.py bit for bit: integer hashing +
// exactly representable float conversions, then vers's own normalize arithmetic (base.rs:99-105:
// sequential f32 dot, sqrt, true division), so any row can be re-generated on the host.
#include "common.hpp"
#include "kmeans.hpp"
#include "util.hip.h"

namespace vers {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float noise_at(uint64_t row_key, uint32_t j) {
  const uint64_t h = mix64(row_key + j);
  const int s = (int)((h & 0xFFFF) + ((h >> 16) & 0xFFFF) + ((h >> 32) & 0xFFFF) + (h >> 48)) - 131070;
  return __fmul_rn((float)s, 1.52587890625e-05f);  // * 2^-16, exact
}

// raw (un-normalised) rows: noise, or centre[row % n_modes] + sigma * noise
__global__ void gen_raw_kernel(float* out, uint64_t n, uint32_t d, uint64_t ld, uint64_t seed, uint64_t start,
                               const float* centres, uint32_t n_modes, float sigma) {
  const uint64_t total = n * ld;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = i / ld;
    const uint32_t j = (uint32_t)(i % ld);
    float v = 0.0f;
    if (j < d) {
      const uint64_t row = start + r;
      const uint64_t rk = mix64(seed + row * 0xD1342543DE82EF95ull);
      v = noise_at(rk, j);
      if (centres) v = __fadd_rn(centres[(row % n_modes) * (uint64_t)ld + j], __fmul_rn(sigma, v));
    }
    out[i] = v;
  }
}

// magnitude per row: sqrt of the sequential dot (one lane per row; L1/L2 absorb the strided walk)
__global__ void row_norm_kernel(const float* rows, uint64_t n, uint32_t d, uint64_t ld, float* mag) {
  const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const float* p = rows + r * ld;
  float acc = 0.0f;
  for (uint32_t j = 0; j < d; ++j) {
    const float a = p[j];
    acc = __fadd_rn(acc, __fmul_rn(a, a));
  }
  mag[r] = __fsqrt_rn(acc);
}

__global__ void row_scale_kernel(float* rows, uint64_t n, uint32_t d, uint64_t ld, const float* mag) {
  const uint64_t total = n * ld;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = i / ld;
    const uint32_t j = (uint32_t)(i % ld);
    const float m = mag[r];
    if (j < d && !(m < 1e-6f)) rows[i] = __fdiv_rn(rows[i], m);
  }
}

static int32_t gen_normalised(float* out, uint64_t n, uint32_t d, uint64_t ld, uint64_t seed, uint64_t start,
                              const float* centres, uint32_t n_modes, float sigma, hipStream_t st) {
  if (n == 0) return VERS_OK;
  DevBuf mag;
  if (int32_t rc = mag.reserve(n * sizeof(float))) return rc;
  uint64_t blocks = (n * ld + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(gen_raw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, n, d, ld, seed, start, centres, n_modes, sigma);
  VERS_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(row_norm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n, d, ld, mag.as<float>());
  VERS_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, n, d, ld, mag.as<float>());
  VERS_HIP_TRY(hipGetLastError());
  VERS_HIP_TRY(hipStreamSynchronize(st));  // mag is freed on return
  return VERS_OK;
}

}  // namespace vers

using namespace vers;

extern "C" int32_t vers_gen_rows_dev(float* out_dev, uint64_t n, uint32_t d, uint64_t ld_floats, uint32_t kind, uint64_t seed,
                                     uint64_t seed_centres, uint32_t n_modes, float sigma, uint64_t start_row, void* stream) {
  if ((n && !out_dev) || d == 0 || ld_floats < d || kind > 1 || (kind == 1 && n_modes == 0))
    return fail(VERS_ERR_INVALID, "vers_gen_rows_dev: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0) return gen_normalised(out_dev, n, d, ld_floats, seed, start_row, nullptr, 0, 0.0f, st);
  DevBuf centres;
  if (int32_t rc = centres.reserve((size_t)n_modes * ld_floats * sizeof(float))) return rc;
  if (int32_t rc = gen_normalised(centres.as<float>(), n_modes, d, ld_floats, seed_centres, 0, nullptr, 0, 0.0f, st)) return rc;
  return gen_normalised(out_dev, n, d, ld_floats, seed, start_row, centres.as<float>(), n_modes, sigma, st);
}
