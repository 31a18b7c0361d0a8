// probe.hip (libvers_hip_test.so) -- TEST HOOK: what the matrix cores do to the certificates' error model, measured.
//
// The three pre-filters (coarse quantiser, list scan, k-means assign) certify their results with a bound that assumes
// "a matrix-core dot product errs like a chain of f32 additions: at most K roundings of at most u = 2^-24 relative each,
// applied to partial sums no larger than sum |a_i b_i|" (gemm.hip.h).  The guide documents that for the f32 MFMA only;
// the production filters run v_mfma_f32_32x32x16_bf16 (x3), v_mfma_f32_32x32x16_f16 (hi + lo) and, on f32 rows,
// v_mfma_f32_16x16x1_4b_f32.  vers_test_mfma runs ONE wave of exactly those instructions over caller-chosen operands --
// subnormal fp16 inputs, maximum magnitudes, cancellation patterns -- accumulating over K the way the kernels do, and
// hands the f32 result back; tests/test_mfma_model_gpu.py compares with the exact rational value.
#include <vector>

#include "../gemm.hip.h"
#include "../util.hip.h"
#include "../../../include/vers_hip_test.h"

namespace vers {

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;  // (prescan.hip.h's operand type; that header defines kernels of the index)

// kind 0: f16 32x32x16 | 1: bf16 32x32x16 | 2: f32 32x32x2 | 3: f32 16x16x1, four blocks
// A [rows][K] row-major, B [K][cols] row-major (rows x cols = 32 x 32; kind 3: 64 x 16), 16-bit operands as bit patterns.
template <int KIND>
__global__ __launch_bounds__(kWave) void mfma_probe_kernel(const void* Av, const void* Bv, uint32_t K, float* C) {
  const int lane = threadIdx.x;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
  if constexpr (KIND == 0 || KIND == 1) {
    const uint16_t* A = reinterpret_cast<const uint16_t*>(Av);
    const uint16_t* B = reinterpret_cast<const uint16_t*>(Bv);
    const int r = lane & 31, kq = 8 * (lane >> 5);
    for (uint32_t k0 = 0; k0 < K; k0 += 16) {
      uint16_t a[8], b[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a[j] = A[(size_t)r * K + k0 + kq + j];
        b[j] = B[(size_t)(k0 + kq + j) * 32 + r];
      }
      if constexpr (KIND == 0) {
        f16x8_t av, bv;
        __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
      } else {
        bf16x8 av, bv;
        __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) C[(size_t)((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[i];
  } else if constexpr (KIND == 2) {
    const float* A = reinterpret_cast<const float*>(Av);
    const float* B = reinterpret_cast<const float*>(Bv);
    const int r = lane & 31, kq = lane >> 5;
    for (uint32_t k0 = 0; k0 < K; k0 += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(size_t)r * K + k0 + kq], B[(size_t)(k0 + kq) * 32 + r], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) C[(size_t)((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[i];
  } else {
    const float* A = reinterpret_cast<const float*>(Av);
    const float* B = reinterpret_cast<const float*>(Bv);
    for (uint32_t k = 0; k < K; ++k)  // block b = lane >> 4: rows 16 b .. 16 b + 15 of A against the same 16 columns of B
      acc = __builtin_amdgcn_mfma_f32_16x16x1f32(A[(size_t)lane * K + k], B[(size_t)k * 16 + (lane & 15)], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) C[(size_t)(16 * (i >> 2) + 4 * (lane >> 4) + (i & 3)) * 16 + (lane & 15)] = acc[i];
  }
}

}  // namespace vers

using namespace vers;

extern "C" int32_t vers_test_mfma(int32_t device, uint32_t kind, const void* A, const void* B, uint32_t K, float* out_C) {
  if (kind > 3 || !A || !B || !out_C || K == 0 || (kind <= 1 && K % 16) || (kind == 2 && K % 2)) return fail(VERS_ERR_INVALID, "vers_test_mfma: bad arguments");
  DeviceGuard g(device);
  const size_t esz = kind <= 1 ? 2 : 4;
  const size_t rows = kind == 3 ? 64 : 32, cols = kind == 3 ? 16 : 32;
  void *dA = nullptr, *dB = nullptr;
  float* dC = nullptr;
  int32_t rc = VERS_OK;
  do {
    if (hipMalloc(&dA, rows * K * esz) != hipSuccess || hipMalloc(&dB, (size_t)K * cols * esz) != hipSuccess ||
        hipMalloc((void**)&dC, rows * cols * sizeof(float)) != hipSuccess) { rc = fail(VERS_ERR_HIP, "vers_test_mfma: allocation failed"); break; }
    if (hipMemcpy(dA, A, rows * K * esz, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(dB, B, (size_t)K * cols * esz, hipMemcpyHostToDevice) != hipSuccess) {
      rc = fail(VERS_ERR_HIP, "vers_test_mfma: upload failed"); break;
    }
    switch (kind) {
      case 0: hipLaunchKernelGGL(mfma_probe_kernel<0>, dim3(1), dim3(kWave), 0, nullptr, dA, dB, K, dC); break;
      case 1: hipLaunchKernelGGL(mfma_probe_kernel<1>, dim3(1), dim3(kWave), 0, nullptr, dA, dB, K, dC); break;
      case 2: hipLaunchKernelGGL(mfma_probe_kernel<2>, dim3(1), dim3(kWave), 0, nullptr, dA, dB, K, dC); break;
      default: hipLaunchKernelGGL(mfma_probe_kernel<3>, dim3(1), dim3(kWave), 0, nullptr, dA, dB, K, dC); break;
    }
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(out_C, dC, rows * cols * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(VERS_ERR_HIP, "vers_test_mfma: launch failed");
  } while (0);
  for (void* p : {dA, dB, (void*)dC})
    if (p) (void)hipFree(p);
  return rc;
}
