// Probe (development aid): a chain of DEPENDENT v_mfma_f32_16x16x1_4b_f32 (same accumulator) with one independent VALU
// instruction between consecutive MFMAs -- does the accumulate chain lose contributions (late-written registers)?
// Two waves per SIMD run the same chain to mimic the scan kernel's occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512) void probe(float* out, int variant) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  float a = 1.0f, b = 1.0f, t = (float)lane, t2 = 3.0f;
  for (int rep = 0; rep < 64; ++rep) {
    if (variant == 0) {        // strictly back to back
      asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0\n\tv_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0\n\tv_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0\n\tv_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0"
                   : "+v"(acc) : "v"(a), "v"(b));
    } else if (variant == 1) { // one VALU instruction between dependent MFMAs
      asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_add_f32 %1, %1, %4\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_add_f32 %1, %1, %4\n\t"
                   "v_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_add_f32 %1, %1, %4\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_add_f32 %1, %1, %4"
                   : "+v"(acc), "+v"(t) : "v"(a), "v"(b), "v"(t2));
    } else if (variant == 3) { // s_nop 0 between
      asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0\n\ts_nop 0\n\tv_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0\n\ts_nop 0\n\tv_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0\n\ts_nop 0\n\tv_mfma_f32_16x16x1_4b_f32 %0, %1, %2, %0"
                   : "+v"(acc) : "v"(a), "v"(b));
    } else if (variant == 4) { // VALU + s_nop 7 between
      asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_add_f32 %1, %1, %4\n\ts_nop 7\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_add_f32 %1, %1, %4\n\ts_nop 7\n\t"
                   "v_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_add_f32 %1, %1, %4\n\ts_nop 7\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_add_f32 %1, %1, %4\n\ts_nop 7"
                   : "+v"(acc), "+v"(t) : "v"(a), "v"(b), "v"(t2));
    } else if (variant == 5) { // VALU + 4 x s_nop 7 (32+ cycles) between
      asm volatile("v_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_add_f32 %1, %1, %4\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_add_f32 %1, %1, %4\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                   "v_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_add_f32 %1, %1, %4\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_add_f32 %1, %1, %4\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                   : "+v"(acc), "+v"(t) : "v"(a), "v"(b), "v"(t2));
    } else {                   // the A operand of every MFMA is produced by the VALU instruction right before it
      asm volatile("v_mov_b32 %1, %2\n\tv_mfma_f32_16x16x1_4b_f32 %0, %1, %3, %0\n\tv_mov_b32 %1, %2\n\tv_mfma_f32_16x16x1_4b_f32 %0, %1, %3, %0\n\t"
                   "v_mov_b32 %1, %2\n\tv_mfma_f32_16x16x1_4b_f32 %0, %1, %3, %0\n\tv_mov_b32 %1, %2\n\tv_mfma_f32_16x16x1_4b_f32 %0, %1, %3, %0"
                   : "+v"(acc), "+v"(t) : "v"(a), "v"(b));
    }
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");
  for (int e = 0; e < 16; ++e) out[(wave * 64 + lane) * 16 + e] = acc[e];
  if (t == -1.0f) out[0] = t;
}
int main() {
  float* d; hipMalloc(&d, 512 * 16 * 4);
  static float h[512 * 16];
  for (int variant = 0; variant < 6; ++variant) {
    probe<<<512, 512>>>(d, variant);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0, by_reg[16] = {0};
    for (int i = 0; i < 512 * 16; ++i) if (h[i] != 256.0f) { ++bad; by_reg[i % 16]++; }
    printf("variant %d (%s): %d of %d results != 256", variant, variant == 0 ? "back to back" : variant == 1 ? "independent VALU between" : variant == 2 ? "A operand written right before" : variant == 3 ? "s_nop 0 between" : variant == 4 ? "VALU + s_nop 7 between" : "VALU + 4 x s_nop 7 between", bad, 512 * 16);
    if (bad) { printf("; per accumulator register:"); for (int e = 0; e < 16; ++e) printf(" %d", by_reg[e]); printf("; lane 0 values:"); for (int e = 0; e < 16; ++e) printf(" %g", h[e]); }
    printf("\n");
  }
  return 0;
}
