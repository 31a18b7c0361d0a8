// wide.hip.h -- key lists WIDER than one key per lane: 4 registers x 64 lanes = 256 ascending keys per wave (element e sits in
// register e / 64 of lane e % 64).  The reference's search_approximate has no cap on top_k (ivfflat.rs:153); the matrix-core list
// scan's candidate lists were one key per lane wide through round 5 (top_k + slack <= 64), wider results went to the ordered
// chains 64 ranks per pass at half the bytes per second.  These networks carry results of up to kWideMaxKp - slack keys through the
// same pre-selection / certificate / exact-finish path (prescan.hip.h WIDE, finish_wide.hip.h).
// Everything here is the bitonic network of scan.hip.h one level up: the stages that cross registers are plain compare-exchanges
// between two registers of a lane, the rest are the 64-lane DPP / permlane stages.
#pragma once
#include "scan.hip.h"

namespace vers {

constexpr int kWideR = 4;                                   // registers of a wide list
constexpr uint32_t kWideKeys = (uint32_t)kWideR * kWave;    // 256 keys
constexpr uint32_t kWideMaxKp = kWideKeys - 24u;            // a compacted candidate buffer leaves >= 24 free slots (prescan.hip.h)

__device__ __forceinline__ void cmpex64(uint64_t& lo, uint64_t& hi) {
  const bool sw = hi < lo;
  const uint64_t a = sw ? hi : lo, b = sw ? lo : hi;
  lo = a; hi = b;
}
// a BITONIC sequence over the NR x 64 elements -> ascending
template <int NR>
__device__ __forceinline__ void wide_bitonic_merge(uint64_t (&k)[NR], int lane) {
#pragma unroll
  for (int dist = NR / 2; dist >= 1; dist >>= 1)
#pragma unroll
    for (int i = 0; i < NR; ++i)
      if ((i & dist) == 0) cmpex64(k[i], k[i + dist]);
#pragma unroll
  for (int r = 0; r < NR; ++r) wave_bitonic_merge64(k[r], lane);
}
// 256 keys in any order (kKeyMax padded) -> ascending.  ROLLED: the per-register rank sorts four lanes per trip (inside a kernel
// with no registers to spare)
template <bool ROLLED = false>
__device__ __forceinline__ void wide_sort(uint64_t (&k)[kWideR], int lane) {
#pragma unroll
  for (int r = 0; r < kWideR; ++r) wave_rank_sort64<ROLLED>(k[r], lane);
  {  // runs of one register -> runs of two: the second run reversed makes the pair bitonic
    uint64_t p[2] = {k[0], lane_rev64(k[1], lane)};
    wide_bitonic_merge<2>(p, lane);
    uint64_t q[2] = {k[2], lane_rev64(k[3], lane)};
    wide_bitonic_merge<2>(q, lane);
    // ... and runs of two -> the whole: (q[0], q[1]) reversed as a 128-key sequence is (rev q[1], rev q[0])
    k[0] = p[0]; k[1] = p[1]; k[2] = lane_rev64(q[1], lane); k[3] = lane_rev64(q[0], lane);
  }
  wide_bitonic_merge<kWideR>(k, lane);
}
// the 256 smallest keys of two ascending wide lists, ascending: one reversed, the element-wise minimum is bitonic
__device__ __forceinline__ void wide_merge_sorted(uint64_t (&list)[kWideR], const uint64_t (&cand)[kWideR], int lane) {
#pragma unroll
  for (int r = 0; r < kWideR; ++r) {
    const uint64_t rv = lane_rev64(cand[kWideR - 1 - r], lane);
    list[r] = list[r] < rv ? list[r] : rv;
  }
  wide_bitonic_merge<kWideR>(list, lane);
}
// element e (wave-uniform) of a wide list.  WHOLE WAVE: the register is picked per lane before one lane of it is read -- under a
// divergent branch the other lanes' copy would be stale.
__device__ __forceinline__ uint64_t wide_get(const uint64_t (&k)[kWideR], uint32_t e) {
  const uint32_t r = e >> 6;
  const uint64_t v = r == 0 ? k[0] : (r == 1 ? k[1] : (r == 2 ? k[2] : k[3]));
  return readlane64(v, (int)(e & 63u));
}

}  // namespace vers
