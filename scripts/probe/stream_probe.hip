// stream_probe.hip -- how fast can ONE launch of the list scan's shape stream the fp16 shadow, as a function of the bytes a CU
// keeps in flight?  (the sharded scan's fixed cost.)
//
// The matrix-core list scan (prescan.hip.h) runs one block of 8 waves per CU, every wave with one 8 KiB step in flight while it
// feeds the previous one to the matrix cores: ~96 KiB in flight per CU.  On one GPU that is enough to hold HBM at its ceiling
// (256 CUs x 96 KiB against ~3 us of loaded latency), but a scan of an eighth of the corpus spends a third of its time with part
// of the chip in hand-out gaps, in the tail, or lent to another batch's small kernels -- and then a CU can go no faster than its
// bytes in flight allow.  This probe isolates that: the same access pattern (a wave = 8 x 1 KiB contiguous loads per step, 16
// v_mfma_f32_32x32x16_f16 per step on what landed), the same dynamic hand-out of whole work items (sizes spread like list
// lengths, longest first), nothing else -- for W waves per block x R steps of ring x B blocks per CU.
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip && ./stream_probe [total MB] [avg item KB]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

struct Item { unsigned long long off; unsigned steps; unsigned pad; };  // steps of 8 KiB

template <int WAVES, int R, int WPE>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void stream_kernel(const char* base, const Item* items, unsigned n_items,
                                                                                                              unsigned* next, float* sink, int gap_us, unsigned unit) {
  __shared__ unsigned s_it;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  f32x16 acc0 = {}, acc1 = {};
  f16x8 bq;
  for (int e = 0; e < 8; ++e) bq[e] = (_Float16)(0.001f * (float)(lane + e));
  for (;;) {
    if (threadIdx.x == 0) s_it = atomicAdd(next, 1u);
    __syncthreads();
    const unsigned it = s_it;
    if (it >= n_items) break;
    const Item im = items[it];
    // the block's waves share the item: wave w walks a contiguous run of its steps
    // (the real kernel hands a wave whole 64-row tiles = 12 steps at d = 768: `unit` = the granularity of the split in steps)
    const unsigned n_units = (im.steps + unit - 1) / unit;
    const unsigned per = (n_units + WAVES - 1) / WAVES * unit;
    const unsigned s0 = min(im.steps, (unsigned)wid * per), s1 = min(im.steps, s0 + per);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(base + im.off), 0, (int)(im.steps * 8192u), 0x00020000);
    u32x4 buf[R][8];
    auto issue = [&](int b, unsigned s) {
#pragma unroll
      for (int i = 0; i < 8; ++i) buf[b][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)lane * 16u, s * 8192u + (unsigned)i * 1024u, 2);
    };
    auto consume = [&](int b) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f16x8 a = __builtin_bit_cast(f16x8, buf[b][i]);
        if (i & 1) { acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq, acc1, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq, acc1, 0, 0, 0); }
        else { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq, acc0, 0, 0, 0); acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq, acc0, 0, 0, 0); }
      }
    };
    const unsigned last = s1 > s0 ? s1 - 1 : s0;
    // prologue: R - 1 steps in flight (clamped re-reads past the end keep every load unconditional)
#pragma unroll
    for (int r = 0; r < R - 1; ++r) issue(r, min(s0 + (unsigned)r, last));
    if (s1 > s0) {
      for (unsigned s = s0; s < s1; s += R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          issue((r + R - 1) % R, min(s + (unsigned)r + R - 1, last));
          if (s + r < s1) consume(r);
        }
      }
    }
    if (gap_us > 0) {  // what stands between two items of a block in the real kernel: write-out, staging, barriers
      __syncthreads();
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)gap_us * 100ull) __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
  }
  float t = 0.0f;
  for (int e = 0; e < 16; ++e) t += acc0[e] + acc1[e];
  if (t == 12345.678f) sink[threadIdx.x] = t;
}

template <int WAVES, int R, int WPE>
void run(const char* name, int blocks_per_cu, const char* base, const Item* d_items, unsigned n_items, unsigned* d_next, float* d_sink, double bytes, int gap_us, int n_cu, unsigned unit) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f, sum = 0.0f;
  const int reps = 12;
  for (int i = 0; i < reps + 2; ++i) {
    CK(hipMemsetAsync(d_next, 0, 4, nullptr));
    CK(hipEventRecord(a, nullptr));
    hipLaunchKernelGGL((stream_kernel<WAVES, R, WPE>), dim3(n_cu * blocks_per_cu), dim3(64 * WAVES), 0, nullptr, base, d_items, n_items, d_next, d_sink, gap_us, unit);
    CK(hipEventRecord(b, nullptr));
    CK(hipEventSynchronize(b));
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, a, b));
    if (i >= 2) { best = std::min(best, ms); sum += ms; }
  }
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, stream_kernel<WAVES, R, WPE>, 64 * WAVES, 0));
  printf("%-34s unit %2u steps  blocks/CU %d (resident %d)  in flight/CU %4d KiB  mean %7.1f us  min %7.1f us  -> %5.2f TB/s (mean)\n", name, unit, blocks_per_cu, occ,
         std::min(occ, blocks_per_cu) * WAVES * (R - 1) * 8, sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
  const double total_mb = argc > 1 ? atof(argv[1]) : 1850.0;   // one rank of eight at cfg3: 1.15M probed rows x 1540 B
  const double avg_kb = argc > 2 ? atof(argv[2]) : 3670.0;     // an average list: 2441 rows x 1540 B
  const int gap_us = argc > 3 ? atoi(argv[3]) : 0;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  // item sizes spread like the list lengths of the bench index (0.3 .. 2.2 x the mean), whole 8 KiB steps, longest first
  std::vector<Item> items;
  unsigned long long off = 0;
  unsigned seed = 12345u;
  while (off < (unsigned long long)(total_mb * 1e6)) {
    seed = seed * 1664525u + 1013904223u;
    const double f = 0.3 + 1.9 * ((seed >> 8) & 0xFFFF) / 65535.0 * ((seed >> 24) & 1 ? 0.6 : 1.0);
    const unsigned steps = std::max(8u, (unsigned)(avg_kb * f * 1000.0 / 8192.0));
    items.push_back(Item{off, steps, 0});
    off += (unsigned long long)steps * 8192ull;
  }
  std::sort(items.begin(), items.end(), [](const Item& x, const Item& y) { return x.steps > y.steps; });
  const double bytes = (double)off;
  char* base; Item* d_items; unsigned* d_next; float* d_sink;
  CK(hipMalloc(&base, off + 65536)); CK(hipMemset(base, 0x11, off + 65536));
  CK(hipMalloc(&d_items, items.size() * sizeof(Item))); CK(hipMemcpy(d_items, items.data(), items.size() * sizeof(Item), hipMemcpyHostToDevice));
  CK(hipMalloc(&d_next, 64)); CK(hipMalloc(&d_sink, 4096 * 4));
  printf("%d CUs; %.2f GB in %zu items (avg %.0f KB, %.1f per CU), longest first; gap between a block's items %d us; ideal at 6.3 TB/s: %.0f us\n", n_cu, bytes / 1e9,
         items.size(), bytes / items.size() / 1e3, (double)items.size() / n_cu, gap_us, bytes / 6.3e12 * 1e6);
  const unsigned n = (unsigned)items.size();
#define RUN(W, R, WPE, B) run<W, R, WPE>(#W " waves x ring " #R " (" #WPE "/SIMD)", B, base, d_items, n, d_next, d_sink, bytes, gap_us, n_cu, unit)
  for (unsigned unit : {1u, 6u, 12u}) {
  RUN(8, 2, 2, 1);    // the list scan today (unit 12: whole tiles per wave)
  }
  unsigned unit = 1;
  RUN(8, 3, 2, 1);
  RUN(8, 4, 2, 1);
  RUN(12, 2, 3, 1);
  RUN(16, 2, 4, 1);
  RUN(16, 3, 4, 1);
  RUN(4, 2, 2, 2);    // two half-size blocks per CU (out of phase), the same bytes in flight as today
  RUN(8, 2, 4, 2);    // two blocks of 8 per CU: twice the bytes
  RUN(8, 3, 4, 2);
  return 0;
}
