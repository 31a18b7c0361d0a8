// spin_residency.hip -- why a software grid barrier cannot be trusted once launches share the chip ("the hang").
//
// Round 2 planned a batch in ONE launch of 64 blocks x 1024 threads (a whole CU each: 16 waves x 128 registers) separated
// by grid barriers that SPIN on a device counter.  A spinning grid is only safe while all of its blocks are resident.
// This probe launches W such grids at once, each on its own stream -- what W host threads (or one thread rotating over W
// streams with a workspace each) did with the library -- with the spin BOUNDED: a block that waits longer than the
// deadline records a timeout and leaves, where the library's kernel would have spun for ever (= a hung GPU: the
// resident blocks of every grid hold the CUs that the others' missing blocks need).  Whole-CU blocks of a terminating
// kernel (the list scan's shape) run in between.
//   hipcc --offload-arch=gfx950 -O2 -o spin_residency spin_residency.hip && ./spin_residency
// prints, per W, launches / launches with a timed-out barrier / longest barrier wait.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr unsigned kBlocks = 64, kThreads = 1024;
constexpr size_t kWholeCuLds = 100 * 1024;  // one block per CU (the library's kernel got there through its registers)

struct Out { unsigned timeouts; unsigned pad; unsigned long long max_wait_ticks; };

// three barrier phases like plan_fused_kernel; the counter is monotonic (never reset), `base` = arrivals before this launch
__global__ __launch_bounds__(kThreads) void spin_grid(unsigned* ctr, unsigned base, unsigned long long deadline_ticks, Out* out) {
  extern __shared__ unsigned lds[];
  lds[threadIdx.x] = threadIdx.x;  // (keeps the allocation)
  for (unsigned phase = 1; phase <= 3; ++phase) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = base + phase * kBlocks;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
      bool late = false;
      while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(16);
        if (__builtin_amdgcn_s_memrealtime() - t0 > deadline_ticks) { late = true; break; }
      }
      const unsigned long long w = __builtin_amdgcn_s_memrealtime() - t0;
      atomicMax(&out->max_wait_ticks, w);
      if (late) atomicAdd(&out->timeouts, 1u);
    }
    __syncthreads();
  }
}
// a terminating kernel whose blocks take a whole CU each for `ticks` (the matrix-core list scan's shape: 256 blocks, 8 waves, 130 KB LDS)
__global__ __launch_bounds__(512) void filler(unsigned long long ticks, unsigned long long stagger) {  // block i runs ticks + i * stagger
  extern __shared__ unsigned lds[];
  lds[threadIdx.x] = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), dur = ticks + blockIdx.x * stagger;
  while (__builtin_amdgcn_s_memrealtime() - t0 < dur) __builtin_amdgcn_s_sleep(32);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 40;
  const unsigned long long deadline = 2000000ull;  // 20 ms: two hundred times the longest honest wait seen with W = 1
  CK(hipFuncSetAttribute((const void*)spin_grid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWholeCuLds));
  CK(hipFuncSetAttribute((const void*)filler, hipFuncAttributeMaxDynamicSharedMemorySize, 130 * 1024));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs; spin grid = %u blocks x %u threads, one block per CU; deadline %.0f ms\n", prop.name, prop.multiProcessorCount,
         kBlocks, kThreads, deadline / 1e5);
  // mode 0: W spin grids on an idle chip | 1: each stream alternates list-scan-shaped launches and spin grids |
  // 2: ONE list-scan-shaped launch holds every CU and its blocks end one by one (1 us apart) while W spin grids, each on its
  //    own stream, are already queued: the CUs that come free are dealt to the waiting grids
  for (int with_filler = 0; with_filler <= 2; ++with_filler)
    for (int W : {1, 2, 3, 4, 5, 6, 8}) {
      std::vector<hipStream_t> st(W);
      std::vector<unsigned*> ctr(W);
      Out* out; CK(hipMalloc(&out, sizeof(Out) * W)); CK(hipMemset(out, 0, sizeof(Out) * W));
      for (int i = 0; i < W; ++i) { CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); CK(hipMalloc(&ctr[i], 64)); CK(hipMemset(ctr[i], 0, 64)); }
      CK(hipDeviceSynchronize());
      unsigned bad_launches = 0;
      hipStream_t st0; CK(hipStreamCreateWithFlags(&st0, hipStreamNonBlocking));
      for (int r = 0; r < rounds; ++r) {
        if (with_filler == 2) hipLaunchKernelGGL(filler, dim3(256), dim3(512), 130 * 1024, st0, 30000ull /* 300 us */, 100ull /* + 1 us per block */);
        for (int i = 0; i < W; ++i) {
          if (with_filler == 1) hipLaunchKernelGGL(filler, dim3(256), dim3(512), 130 * 1024, st[i], 20000ull /* 200 us */, 0ull);
          hipLaunchKernelGGL(spin_grid, dim3(kBlocks), dim3(kThreads), kWholeCuLds, st[i], ctr[i], (unsigned)r * 3u * kBlocks, deadline, out + i);
        }
        CK(hipDeviceSynchronize());
        std::vector<Out> h(W);
        CK(hipMemcpy(h.data(), out, sizeof(Out) * W, hipMemcpyDeviceToHost));
        unsigned t = 0;
        for (auto& o : h) t += o.timeouts;
        if (t) {  // the counters of a timed-out launch are short of their target: start the next round clean
          ++bad_launches;
          for (int i = 0; i < W; ++i) CK(hipMemset(ctr[i], 0, 64));
          std::vector<Out> z(W); for (int i = 0; i < W; ++i) { z[i] = h[i]; z[i].timeouts = 0; }
          CK(hipMemcpy(out, z.data(), sizeof(Out) * W, hipMemcpyHostToDevice));
          // counters restart at 0: the next launches must use base 0 again
          for (int i = 0; i < W; ++i) { /* base is r*3*kBlocks in the launch: realign by adding the missing arrivals */
            unsigned v = (unsigned)(r + 1) * 3u * kBlocks; CK(hipMemcpy(ctr[i], &v, 4, hipMemcpyHostToDevice)); }
        }
      }
      std::vector<Out> h(W); CK(hipMemcpy(h.data(), out, sizeof(Out) * W, hipMemcpyDeviceToHost));
      unsigned long long mw = 0; for (auto& o : h) mw = o.max_wait_ticks > mw ? o.max_wait_ticks : mw;
      printf("%s  W=%d concurrent spin grids (%3d whole CUs wanted): %d rounds, %u with a timed-out barrier (= a hang without the deadline); longest barrier wait %.1f us\n",
             with_filler == 2 ? "queued behind ONE draining list scan     " : with_filler ? "with whole-CU list-scan blocks in between" : "spin grids only                          ", W, W * (int)kBlocks, rounds, bad_launches, mw / 100.0);
      CK(hipStreamDestroy(st0));
      for (int i = 0; i < W; ++i) { CK(hipStreamDestroy(st[i])); CK(hipFree(ctr[i])); }
      CK(hipFree(out));
    }
  return 0;
}
