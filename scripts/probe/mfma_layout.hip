// Probe of the v_mfma_f32_16x16x1_4b_f32 operand/result layout on gfx950 (development aid, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void probe(float* out) {
  const int lane = threadIdx.x;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
  // A = 1000 + lane (row tag), B = 1 for every lane with (lane & 15) == sel, one sel at a time -> D = A-tag of the row, in column sel
  for (int sel = 0; sel < 16; ++sel) {
    f32x16 a2;
    for (int e = 0; e < 16; ++e) a2[e] = 0.0f;
    a2 = __builtin_amdgcn_mfma_f32_16x16x1f32((float)(1000 + lane), (lane & 15) == sel ? 1.0f : 0.0f, a2, 0, 0, 0);
    for (int e = 0; e < 16; ++e) out[(sel * 64 + lane) * 16 + e] = a2[e];
  }
}
int main() {
  float* d; hipMalloc(&d, 16 * 64 * 16 * 4);
  probe<<<1, 64>>>(d);
  static float h[16 * 64 * 16];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // for B column sel, which lanes/regs are non-zero and what A tag do they carry
  int bad = 0;
  for (int sel = 0; sel < 16; ++sel)
    for (int lane = 0; lane < 64; ++lane)
      for (int e = 0; e < 16; ++e) {
        const float v = h[(sel * 64 + lane) * 16 + e];
        const int want_row = 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);
        const float want = ((lane & 15) == sel) ? (float)(1000 + want_row) : 0.0f;
        if (v != want) { if (bad < 10) printf("sel %d lane %d e %d: got %g want %g\n", sel, lane, e, v, want); ++bad; }
      }
  printf("layout hypothesis (col = lane&15, row = 16*(e>>2) + 4*(lane>>4) + (e&3)): %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  for (int e = 0; e < 16; ++e) printf("lane 17 e %d -> %g\n", e, h[(1 * 64 + 17) * 16 + e]);
  return 0;
}
