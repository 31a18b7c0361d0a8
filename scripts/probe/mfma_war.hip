// Probe (development aid): does a VALU write to the B operand register of an MFMA that was just issued corrupt the
// MFMA's late passes on gfx950?  v_mfma_f32_16x16x1_4b_f32 D = A(lane) x B(lane & 15) ; then v_mov overwrites B.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void probe(float* out, int with_nop) {
  const int lane = threadIdx.x;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  float a = 1.0f, b = 2.0f;
  f32x16 res;
  float b2 = 3.0f;
  // a chain of dependent MFMAs (same accumulator), the last one with its own B register, overwritten right behind it
  if (with_nop) {
    asm volatile("s_nop 4\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %1\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, %4, %0\n\t"
                 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %4, 0x42c80000\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                 : "=&v"(res), "+v"(acc), "+v"(a), "+v"(b), "+v"(b2));
  } else {
    asm volatile("s_nop 4\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %1\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, %3, %0\n\tv_mfma_f32_16x16x1_4b_f32 %0, %2, %4, %0\n\t"
                 "v_mov_b32 %4, 0x42c80000\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                 : "=&v"(res), "+v"(acc), "+v"(a), "+v"(b), "+v"(b2));
  }
  acc = res;
  for (int e = 0; e < 16; ++e) out[lane * 16 + e] = acc[e];
  out[64 * 16 + lane] = b;
}
int main() {
  float* d; hipMalloc(&d, (64 * 16 + 64) * 4);
  static float h[64 * 16 + 64];
  for (int with_nop = 1; with_nop >= 0; --with_nop) {
    probe<<<1, 64>>>(d, with_nop);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0, first = -1;
    for (int i = 0; i < 64 * 16; ++i) if (h[i] != 9.0f) { if (first < 0) first = i; ++bad; }
    printf("%s: %d of 1024 results != 9.0 (= 2 + 2 + 2 + 3)", with_nop ? "nops before the overwrite" : "overwrite right behind the MFMA", bad);
    if (bad) printf(" (first: lane %d reg %d = %g)", first / 16, first % 16, h[first]);
    printf("\n");
    if (bad) { int by_reg[16] = {0}; for (int i = 0; i < 1024; ++i) if (h[i] != 9.0f) by_reg[i % 16]++; for (int e = 0; e < 16; ++e) printf("%d ", by_reg[e]); printf(" <- wrong results per accumulator register\n"); }
  }
  return 0;
}
