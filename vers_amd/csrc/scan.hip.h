// scan.hip.h -- the exact-order distance engine and wave top-k shared by every
// kernel of the path (flat scan, coarse quantiser, inverted-list scan, k-means
// assign).  gfx950 only.
//
// Arithmetic contract (the reason this file exists): a distance is
//     acc = 0; for j in 0..d: t = x[j] - q[j]; acc = acc + t*t
// evaluated strictly left to right in f32 with separately rounded multiply and
// add (vers base.rs:119-126; Rust never contracts or re-associates).  Each
// LANE owns one corpus row and walks its columns in order, so the result is
// bit-identical to the reference; the matrix is stored in HBM in lane-transposed
// 64-row tiles (below) so that every wave-level load is one contiguous 1 KiB read
// that lands directly in the lanes that own the rows -- no LDS, no shuffles.
#pragma once
#include <type_traits>
#include "common.hpp"

// never contract a*b+c: the reference rounds the product and the sum separately
#pragma clang fp contract(off)

namespace vers {

typedef __attribute__((address_space(4))) const float cfloat_as4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// ---- order-preserving key <-> f32 -------------------------------------------
__device__ __forceinline__ uint32_t f32_to_order_bits(float x) {
  uint32_t b = __float_as_uint(x);
  uint32_t k = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return (x != x) ? 0xFFFFFFFFu : k;  // every NaN sorts last
}
__host__ __device__ __forceinline__ uint32_t order_bits_to_f32_bits(uint32_t k) {
  return (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
}
__device__ __forceinline__ uint64_t make_key(float dist, uint32_t seq) {
  return ((uint64_t)f32_to_order_bits(dist) << 32) | seq;
}

// ---- wave-level helpers ------------------------------------------------------
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane /*uniform*/) {
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, lane);
  uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), lane);
  return ((uint64_t)hi << 32) | lo;
}
// a value that is the same in every lane, moved to the scalar file
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
// value of lane-1 (lane 0 receives 0): DPP wave_shr:1, a plain VALU move -- no LDS round trip
__device__ __forceinline__ uint64_t shift_up1_64(uint64_t v) {
  const uint32_t lo = __builtin_amdgcn_update_dpp(0u, (uint32_t)v, 0x138, 0xf, 0xf, false);
  const uint32_t hi = __builtin_amdgcn_update_dpp(0u, (uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
  return ((uint64_t)hi << 32) | lo;
}

// Sorted list of the 64 smallest keys seen so far, one per lane (ascending by
// lane).  `cand` is one candidate per lane (kKeyMax = none).  Only candidates
// below min(current k-th key, bound) are inserted; keys are unique (seq differs), so
// the order is total and the result equals a stable sort by (dist, seq).
// `bound` is any key known to have at least k keys at or below it among ALL candidates of the
// final merge (a finished work item's k-th key, shared through memory): larger keys can never
// reach the final top-k, so dropping them changes nothing in the result.
__device__ __forceinline__ void wave_topk_update(uint64_t& list, uint32_t k, uint64_t cand, uint64_t bound) {
  uint64_t thr = readlane64(list, (int)k - 1);
  thr = thr < bound ? thr : bound;
  uint64_t m = __ballot(cand < thr);
  while (m) {
    int src = __ffsll((unsigned long long)m) - 1;
    m &= m - 1;
    uint64_t x = readlane64(cand, src);
    if (x < thr) {
      uint64_t prev = shift_up1_64(list);
      uint64_t mx = prev > x ? prev : x;
      list = x < list ? mx : list;
      uint64_t kth = readlane64(list, (int)k - 1);
      thr = kth < thr ? kth : thr;
    }
  }
}

// Minimum of a u32 over the 64 lanes (DPP row shifts + row broadcasts, no LDS); uniform result.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
  const int id = (int)0xFFFFFFFFu;  // identity for lanes a shift does not reach
  uint32_t y;
  y = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x111, 0xf, 0xf, false); x = x < y ? x : y;  // row_shr:1
  y = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x112, 0xf, 0xf, false); x = x < y ? x : y;  // row_shr:2
  y = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x114, 0xf, 0xf, false); x = x < y ? x : y;  // row_shr:4
  y = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x118, 0xf, 0xf, false); x = x < y ? x : y;  // row_shr:8: lane 15 of a row = row min
  y = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x142, 0xa, 0xf, false); x = x < y ? x : y;  // row_bcast:15 into rows 1, 3
  y = (uint32_t)__builtin_amdgcn_update_dpp(id, (int)x, 0x143, 0xc, 0xf, false); x = x < y ? x : y;  // row_bcast:31 into rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// Sum / maximum of a u32 over the 64 lanes, same DPP pattern (row-wise inclusive scan by doubling, then row broadcasts).
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8: lane 15 of a row = row sum
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1, 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
// Inclusive prefix sum of a u32 over the 64 lanes: the same six DPP steps (lane i ends with x_0 + ... + x_i) -- plain VALU moves
// where __shfl_up is a ds_bpermute round trip per step.
__device__ __forceinline__ uint32_t wave_incl_u32(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);  // row_shr:8: inclusive inside each row of 16
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15: row 0's total into row 1, row 2's into row 3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31: the first half's total into rows 2, 3
  return x;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
  uint32_t y;
  y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); x = x > y ? x : y;
  y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false); x = x > y ? x : y;
  y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false); x = x > y ? x : y;
  y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false); x = x > y ? x : y;
  y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); x = x > y ? x : y;
  y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); x = x > y ? x : y;
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// First fold of a work item: the list is EMPTY, so all 64 candidates would pass the threshold and be
// inserted one by one.  Instead pull out the k smallest directly (k rounds of: wave minimum of the
// distance bits, then of the sequence among the ties) and write them into lanes 0..k-1 in order.
// Same result as 64 ordered inserts at about a third of the instructions for k <= 16.
__device__ __forceinline__ void wave_topk_fill(uint64_t& list, uint32_t k, uint64_t cand, int lane) {
  for (uint32_t j = 0; j < k; ++j) {
    const uint32_t hi = (uint32_t)(cand >> 32), lo = (uint32_t)cand;
    if (__ballot(cand != kKeyMax) == 0) break;  // nothing left (a NaN key has all-ones distance bits but a real seq)
    const uint32_t m = wave_min_u32(hi);
    // (ties on the distance bits are rare: with ONE lane at the minimum its sequence number is read off that lane; the second
    // reduction over the tied lanes is a chain of seven dependent DPP steps)
    const uint64_t tie = __ballot(hi == m);
    const uint32_t m2 = (tie & (tie - 1)) == 0 ? (uint32_t)__builtin_amdgcn_readlane((int)lo, __ffsll((unsigned long long)tie) - 1)
                                                : wave_min_u32(hi == m ? lo : 0xFFFFFFFFu);
    const uint64_t x = ((uint64_t)m << 32) | m2;
    if (lane == (int)j) list = x;
    if (cand == x) cand = kKeyMax;
  }
}

// ---- sorting network over the 64 lanes of a wave (ds_bpermute shuffles, no LDS memory) ------------------------
__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int m) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, m, kWave), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m, kWave);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_idx64(uint64_t v, int src) {
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, kWave), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, kWave);
  return ((uint64_t)hi << 32) | lo;
}
// Lane l <- lane l ^ J of the same wave, on the VALU: DPP inside a row of 16 (quad_perm for J = 1, 2; the two row shifts by 4
// and a select; row_ror:8), v_permlane16_swap / v_permlane32_swap (gfx950) across rows and halves.  ~2-6 instructions per 32 bits
// where __shfl_xor is a ds_bpermute: a round trip through the LDS crossbar, ~120 cycles that the six DEPENDENT stages of a merge
// network cannot overlap -- a 64-lane merge was ~1500 cycles, 0.6 us of every fold of two sorted lists (measured on the single
// query's lone finish block, round 5).  Whole wave (every lane active).
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
template <int J>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t x, int lane) {
  static_assert(J == 1 || J == 2 || J == 4 || J == 8 || J == 16 || J == 32, "one bit of the lane index");
#ifdef VERS_LANE_NET_LDS  // (A/B builds, scripts/build_variant.sh: the ds_bpermute route of rounds 1-4)
  return (uint32_t)__shfl_xor((int)x, J, kWave);
#endif
  if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xf, 0xf, false);       // quad_perm [1,0,3,2]
  else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
  else if constexpr (J == 4) {
    const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x104, 0xf, 0xf, false);  // row_shl:4: lane i <- i + 4
    const uint32_t dn = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x114, 0xf, 0xf, false);  // row_shr:4: lane i <- i - 4
    return (lane & 4) ? dn : up;
  } else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x128, 0xf, 0xf, false);  // row_ror:8
  else if constexpr (J == 16) {
    const u32x2_t r = __builtin_amdgcn_permlane16_swap(x, x, false, false);  // {rows [0,0,2,2], rows [1,1,3,3]} of x
    return (lane & 16) ? r[0] : r[1];
  } else {
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(x, x, false, false);  // {halves [lo,lo], halves [hi,hi]}
    return (lane & 32) ? r[0] : r[1];
  }
}
template <int J>
__device__ __forceinline__ uint64_t lane_xor64(uint64_t v, int lane) {
  return ((uint64_t)lane_xor_u32<J>((uint32_t)(v >> 32), lane) << 32) | lane_xor_u32<J>((uint32_t)v, lane);
}
// lane l <- lane 63 - l: row_mirror (15 - i inside a row), then the rows and the halves swapped
__device__ __forceinline__ uint64_t lane_rev64(uint64_t v, int lane) {
#ifdef VERS_LANE_NET_LDS
  return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), kWave - 1 - lane, kWave) << 32) | (uint32_t)__shfl((int)(uint32_t)v, kWave - 1 - lane, kWave);
#endif
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, 0x140, 0xf, 0xf, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), 0x140, 0xf, 0xf, false);
  return lane_xor64<32>(lane_xor64<16>(((uint64_t)hi << 32) | lo, lane), lane);
}
template <int J>
__device__ __forceinline__ void bitonic_stage64(uint64_t& key, int lane, bool up) {
  const uint64_t o = lane_xor64<J>(key, lane);
  const bool lower = (lane & J) == 0;
  key = ((lower == up) == (key < o)) ? key : o;  // (up: the lower lane keeps the minimum, the upper lane the maximum)
}
// the last stage of a bitonic sort: a bitonic sequence over the 64 lanes -> ascending
__device__ __forceinline__ void wave_bitonic_merge64(uint64_t& key, int lane) {
  bitonic_stage64<32>(key, lane, true);
  bitonic_stage64<16>(key, lane, true);
  bitonic_stage64<8>(key, lane, true);
  bitonic_stage64<4>(key, lane, true);
  bitonic_stage64<2>(key, lane, true);
  bitonic_stage64<1>(key, lane, true);
}
__device__ __forceinline__ void wave_bitonic_sort64(uint64_t& key, int lane) {
  auto block = [&](auto ktag) {  // the k-block's stages j = k / 2 .. 1; direction by the lane's k-block (the last one: all ascending)
    constexpr int K = decltype(ktag)::value;
    const bool up = (lane & K) == 0 || K == kWave;
    if constexpr (K >= 64) bitonic_stage64<32>(key, lane, up);
    if constexpr (K >= 32) bitonic_stage64<16>(key, lane, up);
    if constexpr (K >= 16) bitonic_stage64<8>(key, lane, up);
    if constexpr (K >= 8) bitonic_stage64<4>(key, lane, up);
    if constexpr (K >= 4) bitonic_stage64<2>(key, lane, up);
    bitonic_stage64<1>(key, lane, up);
  };
  block(std::integral_constant<int, 2>{});
  block(std::integral_constant<int, 4>{});
  block(std::integral_constant<int, 8>{});
  block(std::integral_constant<int, 16>{});
  block(std::integral_constant<int, 32>{});
  block(std::integral_constant<int, 64>{});
}
// One key per lane -> ascending over the lanes, by RANK COUNTING: every lane counts the lanes whose DISTANCE BITS lie below its
// own (64 scalar reads of a lane, a compare and an add each) and sends its key to the lane of that rank (ds_permute): ~200 VALU
// instructions and two LDS-crossbar round trips against the bitonic network's 21 dependent stages of two ds_bpermute round trips
// each.  Two keys with the same distance bits would claim the same rank: a lane id sent along the same route first tells
// whether every lane is somebody's destination; if not (rare: equal f32 distances inside one 64-key set) the network, which
// orders by (bits, sequence number), does the job.  The kKeyMax lanes take the ranks behind the others in lane order.
template <bool ROLLED = false>  // ROLLED: the counting loop four lanes per trip (inside a kernel that has no registers to spare)
__device__ __forceinline__ void wave_rank_sort64(uint64_t& key, int lane) {
  const uint32_t hi = (uint32_t)(key >> 32);
  uint32_t rank = 0;
  if constexpr (ROLLED) {
#pragma unroll 4
    for (int l = 0; l < kWave; ++l) rank += (uint32_t)__builtin_amdgcn_readlane((int)hi, l) < hi ? 1u : 0u;
  } else {
#pragma unroll
    for (int l = 0; l < kWave; ++l) rank += (uint32_t)__builtin_amdgcn_readlane((int)hi, l) < hi ? 1u : 0u;
  }
  const uint64_t vm = __ballot(key != kKeyMax);
  if (key == kKeyMax) rank = (uint32_t)__popcll(vm) + (uint32_t)__popcll(~vm & ((1ull << lane) - 1ull));
  const int got = __builtin_amdgcn_ds_permute((int)(rank << 2), lane + 1);
  if (__ballot(got == 0) != 0) {  // (wave-uniform)
    wave_bitonic_sort64(key, lane);
    return;
  }
  const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)(uint32_t)key);
  const uint32_t h2 = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)hi);
  key = ((uint64_t)h2 << 32) | lo;
}
// Two ascending 64-lane key lists (kKeyMax padded) -> the 64 smallest keys of their union, ascending: one list reversed, the
// lane-wise minimum is a bitonic sequence, six compare-exchange stages sort it.  (An ordered insert per key -- wave_topk_update
// -- is ~130 cycles per key that passes the threshold: 34 of them for the first slot merged into an empty list.)
__device__ __forceinline__ void wave_merge_sorted64(uint64_t& list, uint64_t cand, int lane) {
  if (__ballot(cand != kKeyMax) == 0) return;  // nothing in it (wave-uniform)
  const uint64_t rev = lane_rev64(cand, lane);
  list = list < rev ? list : rev;
  wave_bitonic_merge64(list, lane);
}
// ---- HBM layout: lane-transposed tiles ---------------------------------------------------------
// A scanned matrix (corpus lists, centroids) is stored in tiles of 64 rows.  Inside tile t
// (base = t*64*ld floats) element (r, j) lives at ((j/4)*64 + r)*4 + (j%4): for each group of 4
// columns the 64 rows' float4s are contiguous (1 KiB).  One `buffer_load_dwordx4` per wave is then
// a single contiguous 1 KiB read AND delivers to lane r exactly row r's 4 columns: the row-per-lane
// operand layout the ordered f32 chain needs, with no LDS transpose and perfectly sequential HBM
// streaming (a wave walks its tile front to back).  ld is a multiple of kColAlign (zero padded).
__host__ __device__ __forceinline__ uint64_t blocked_index(uint64_t row, uint32_t col, uint32_t ld) {
  return (row >> 6) * 64ull * ld + ((uint64_t)(col >> 2) * 64 + (row & 63)) * 4 + (col & 3);
}

// Cache policy of a tile load: rows that a launch streams once (corpus / inverted lists) are loaded non-temporal
// (aux = 2) -- same-box A/B: cfg2 flat scan 95.9 -> 91-94 us, single-query list scan 62.2 -> 57.9 us; matrices that
// every work item re-reads (centroids) keep the default policy so that they stay in L2.  Src::kStreamOnce selects.
template <bool STREAM_ONCE>
constexpr int tile_aux() { return STREAM_ONCE ? 2 : 0; }

struct TileLoader {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t lane_off;    // lane * 16 bytes
  uint32_t tile_bytes;  // 64 * ld * 4

  __device__ __forceinline__ void init(const float* base, uint64_t item_bytes, uint32_t ld, int lane) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(uint32_t)item_bytes, 0x00020000);
    lane_off = (uint32_t)lane * 16u;
    tile_bytes = ld * 256u;
  }
  // the kLoads float4 loads of (tile, chunk c): everything but the lane offset is wave-uniform -> soffset
  template <int AUX>
  __device__ __forceinline__ void issue(u32x4 (&r)[kLoads], uint32_t tile, uint32_t c) const {
    const uint32_t soff = tile * tile_bytes + c * (kLoads * 1024u);
#pragma unroll
    for (int i = 0; i < kLoads; ++i)
      r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, soff + (uint32_t)i * 1024u, AUX);
  }
};

// (x[H], x[H]) - (q.x, q.y) and (x[H], x[H]) * (q.x, q.y): one v_pk_*_f32 whose first operand is a
// natural register pair of the loaded float4 with element H broadcast by op_sel, and whose other
// operand is a register pair holding the same column of two queries.  Register-only asm: nothing
// for the compiler to count or wait for.  IEEE: x - q is computed as x + (-q), exact same result.
template <int H>
__device__ __forceinline__ f32x2 pk_bcast_sub(f32x2 x, f32x2 q) {
  f32x2 t;
  if (H == 0) asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(x), "v"(q));
  else asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(x), "v"(q));
  return t;
}
template <int H>
__device__ __forceinline__ f32x2 pk_bcast_mul(f32x2 x, f32x2 q) {
  f32x2 t;
  if (H == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(x), "v"(q));
  else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(t) : "v"(x), "v"(q));
  return t;
}

// Query operands are wave-uniform.  QG == 1: the query comes in through the scalar path (one
// vector shared by every wave: scalar-cache resident).  QG > 1: the item's queries are stored
// INTERLEAVED, qb[col * QG + qi], so that adjacent pairs feed v_pk_*_f32: two queries per VALU
// instruction, each still its own strictly ordered f32 chain.  NP = live query PAIRS (dead pairs
// cost nothing).  `qb` is a global pointer for QG == 1 and an LDS pointer for QG > 1.
template <int QG, int NP, int METRIC>
__device__ __forceinline__ void tile_chunk_compute(f32x2 (&acc)[(QG + 1) / 2], const u32x4 (&r)[kLoads],
                                                   const float* qb, uint32_t c) {
  if constexpr (QG == 1) {
    cfloat_as4* qs = (cfloat_as4*)(qb + c * kChunk);
    float a = acc[0][0];
#pragma unroll
    for (int i = 0; i < kLoads; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float xv = __uint_as_float(r[i][u]);
        const float sv = qs[i * 4 + u];
        if (METRIC == 0) {
          const float t = __fsub_rn(xv, sv);
          a = __fadd_rn(a, __fmul_rn(t, t));
        } else {
          a = __fadd_rn(a, __fmul_rn(xv, sv));
        }
      }
    }
    acc[0][0] = a;
  } else {
    // Batched: the work item's interleaved query block [col][QG] sits in LDS (staged once per block
    // item by the four waves that share it).  A wave-uniform ds_read_b128 broadcasts one column of 4
    // queries; LDS returns in order, so the compiler pipelines the reads under counted lgkmcnt
    // waits -- unlike scalar loads, whose out-of-order return forces lgkmcnt(0) at every use and
    // whose re-fetch per tile went all the way to HBM.
    static_assert(NP % 2 == 0, "query pairs are consumed two at a time (one ds_read_b128)");
    const f32x4* ql = reinterpret_cast<const f32x4*>(qb) + (size_t)c * kChunk * (QG / 4);
#pragma unroll
    for (int i = 0; i < kLoads; ++i) {
      // the loaded float4 as two natural register pairs; one column is broadcast to both halves of
      // the packed op by op_sel (see pk_bcast_*), so no {x, x} pair is ever materialised
      const f32x2 xlo = {__uint_as_float(r[i][0]), __uint_as_float(r[i][1])};
      const f32x2 xhi = {__uint_as_float(r[i][2]), __uint_as_float(r[i][3])};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x2 xp = u < 2 ? xlo : xhi;
#pragma unroll
        for (int h = 0; h < NP / 2; ++h) {
          const f32x4 q4 = ql[(i * 4 + u) * (QG / 4) + h];
          const f32x2 q01 = {q4[0], q4[1]}, q23 = {q4[2], q4[3]};
          if (METRIC == 0) {
            const f32x2 t0 = (u & 1) ? pk_bcast_sub<1>(xp, q01) : pk_bcast_sub<0>(xp, q01);
            const f32x2 t1 = (u & 1) ? pk_bcast_sub<1>(xp, q23) : pk_bcast_sub<0>(xp, q23);
            acc[2 * h] = acc[2 * h] + t0 * t0;
            acc[2 * h + 1] = acc[2 * h + 1] + t1 * t1;
          } else {
            const f32x2 t0 = (u & 1) ? pk_bcast_mul<1>(xp, q01) : pk_bcast_mul<0>(xp, q01);
            const f32x2 t1 = (u & 1) ? pk_bcast_mul<1>(xp, q23) : pk_bcast_mul<0>(xp, q23);
            acc[2 * h] = acc[2 * h] + t0;
            acc[2 * h + 1] = acc[2 * h + 1] + t1;
          }
        }
      }
    }
  }
}

// Batched math over TWO row tiles at once (lane r owns row r of tile A and row r of tile B): every
// query operand read from LDS feeds both, which halves the LDS return-bus traffic per packed
// instruction -- the bus (128 B/clk/CU, 1 KiB per broadcast ds_read_b128) is what bounds the
// batched math, not the VALU.
template <int QG, int NP, int METRIC>
__device__ __forceinline__ void tile_chunk_compute2(f32x2 (&accA)[QG / 2], f32x2 (&accB)[QG / 2],
                                                    const u32x4 (&r)[2 * kLoads], const float* qb, uint32_t c) {
  const float* ql = qb + (size_t)c * kChunk * QG;
  auto pair_step = [&](f32x2& aA, f32x2& aB, f32x2 xa, f32x2 xb, f32x2 q, int odd) {
    if (METRIC == 0) {
      const f32x2 a = odd ? pk_bcast_sub<1>(xa, q) : pk_bcast_sub<0>(xa, q);
      const f32x2 b = odd ? pk_bcast_sub<1>(xb, q) : pk_bcast_sub<0>(xb, q);
      aA = aA + a * a;
      aB = aB + b * b;
    } else {
      aA = aA + (odd ? pk_bcast_mul<1>(xa, q) : pk_bcast_mul<0>(xa, q));
      aB = aB + (odd ? pk_bcast_mul<1>(xb, q) : pk_bcast_mul<0>(xb, q));
    }
  };
#pragma unroll
  for (int i = 0; i < kLoads; ++i) {
    const f32x2 aLo = {__uint_as_float(r[i][0]), __uint_as_float(r[i][1])};
    const f32x2 aHi = {__uint_as_float(r[i][2]), __uint_as_float(r[i][3])};
    const f32x2 bLo = {__uint_as_float(r[kLoads + i][0]), __uint_as_float(r[kLoads + i][1])};
    const f32x2 bHi = {__uint_as_float(r[kLoads + i][2]), __uint_as_float(r[kLoads + i][3])};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x2 xa = u < 2 ? aLo : aHi, xb = u < 2 ? bLo : bHi;
      const float* qc = ql + (i * 4 + u) * QG;  // this column's QG query values (wave-uniform LDS address)
#pragma unroll
      for (int h = 0; h < NP / 2; ++h) {  // two query pairs per ds_read_b128
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(qc + 4 * h);
        pair_step(accA[2 * h], accB[2 * h], xa, xb, f32x2{q4[0], q4[1]}, u & 1);
        pair_step(accA[2 * h + 1], accB[2 * h + 1], xa, xb, f32x2{q4[2], q4[3]}, u & 1);
      }
      if constexpr (NP % 2 == 1) {  // odd live pair count: the last pair alone (ds_read_b64)
        const f32x2 q2 = *reinterpret_cast<const f32x2*>(qc + 2 * (NP - 1));
        pair_step(accA[NP - 1], accB[NP - 1], xa, xb, q2, u & 1);
      }
    }
  }
}

// ---- the scan kernel -----------------------------------------------------------
// A work item = a run of rows scored against up to QG queries by ONE wave, which
// keeps a sorted top-k list per query across the item's tiles and writes one
// partial slot (k ascending keys) per query.  `Src` maps an item index to an
// ItemView (+ lazily to the per-query seq base and output slot, which are only
// needed outside the inner loop and would otherwise sit in SGPRs); the grid is
// persistent-style (waves stride over items) so that the item count may live in
// device memory (planned on the device, no host sync).
//
// Src interface (all arguments wave-uniform):
//   uint32_t n_items() const                 QG > 1: a multiple of 4, quads share a query block
//   void     get(it, ItemView<QG>&) const    nrows == 0 marks a padding item
//   uint32_t seq_base(it, qi) const          seq of the item's first row for query qi
//   uint64_t* out(it, qi) const              partial slot (k keys) for query qi
//   static constexpr bool kSeqIds            seq = seq_ids(it)[row] instead of seq_base + row
//   static constexpr bool kStreamOnce        rows are read once per launch: non-temporal tile loads
//   const uint32_t* seq_ids(it) const
//   uint32_t bound_slot(it, qi) const        index into ScanParams::bounds of the merge group of (it, qi)
template <int QG>
struct ItemView {
  const float* rows;   // first tile of the item (blocked layout, 64-row aligned)
  uint32_t nrows;      // rows in the item
  uint32_t nq;         // live queries (<= QG)
  const float* qb;     // QG == 1: the query; else the item's interleaved query block [col][QG]
  uint32_t row0 = 0;   // inverted-list items: the item's first row inside its list (sequence numbers; set by IvfSrc only)
};                     // (zero padded to n_chunks*64 columns; dead query slots are zeros)

struct ScanParams {
  uint32_t ld;        // columns of the blocked matrix (multiple of kColAlign)
  uint32_t n_chunks;  // ld / kChunk (even)
  uint32_t k;         // keys kept per query (<= 64)
  uint32_t* status;   // device word: bit0 = NaN seen
  uint64_t* bounds;   // nullable: shared pruning bound per merge group (Src::bound_slot), kKeyMax initialised
  const uint64_t* lower;  // nullable: per merge group (Src::bound_slot) the last key of the PREVIOUS pass -- keys <= it are dropped.
                          // Results wider than one key per lane (top_k or nprobe > 64) are produced 64 ranks per pass: keys are
                          // unique and totally ordered, so pass p holds exactly ranks 64p .. 64p+63 of the full order.
  uint32_t debug;     // diagnosis only (option "scan_debug"): 1 skip top-k, 2 skip math, 4 one query column, 16 stamp phases
  uint32_t* next_quad;  // batched kernels: device counter for dynamic quad hand-out (zeroed per launch) or nullptr
  unsigned long long* stamps;  // debug & 16: [0] cycles waiting for loads, [1] math, [2] top-k fold, [3] item setup, [4] waves
};

// One work item, NP live query pairs (QG == 1: NP == 1).
template <int QG, int NP, int METRIC, class Src>
__device__ __forceinline__ void scan_item(const Src& src, const ScanParams& p, uint32_t it, const ItemView<QG>& v, int lane,
                                          bool& nan_seen) {
  uint64_t list[QG], bound[QG];
#pragma unroll
  for (int qi = 0; qi < QG; ++qi) {
    list[qi] = kKeyMax;
    bound[qi] = kKeyMax;
    if (p.bounds != nullptr && qi < 2 * NP && qi < (int)v.nq)  // relaxed agent-scope read: a stale value only prunes less
      bound[qi] = uniform64(__hip_atomic_load(p.bounds + src.bound_slot(it, qi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  uint64_t vlower = 0;  // lane qi: query qi's exclusive lower bound (0 = none: no key is 0)
  if (p.lower != nullptr && lane < QG && lane < (int)v.nq) vlower = p.lower[src.bound_slot(it, lane)];
  const uint32_t n_tiles = (v.nrows + kWave - 1) / kWave;
  TileLoader L;
  L.init(v.rows, (uint64_t)n_tiles * kWave * p.ld * 4u, p.ld, lane);
  f32x2 acc[(QG + 1) / 2];
#pragma unroll
  for (int p2 = 0; p2 < (QG + 1) / 2; ++p2) acc[p2] = f32x2{0.0f, 0.0f};
  uint32_t sid = 0;
  if (Src::kSeqIds && lane < (int)v.nrows) sid = src.seq_ids(it)[lane];  // older than every prefetch below
  // Per-query item constants (sequence base, output slot) are fetched ONCE, lane qi holding query qi's,
  // and read back with v_readlane: a memory load inside the streaming loop would be waited for with
  // vmcnt(0) and drain the whole prefetch ring at every tile boundary.
  uint32_t vseq = 0;
  uint64_t vout = 0;
  if (lane < QG && lane < (int)v.nq) {
    vseq = Src::kSeqIds ? 0u : src.seq_base(it, lane);
    vout = (uint64_t)src.out(it, lane);
  }

  // end of a tile: fold its 64 candidates into the per-query lists
  auto tile_done = [&](uint32_t t) {
    const uint32_t row = t * kWave + lane;
    // kSeqIds: rows whose id is 0xFFFFFFFF are unused storage slack, not vectors
    const bool valid = row < v.nrows && (!Src::kSeqIds || sid != 0xFFFFFFFFu);
#pragma unroll
    for (int qi = 0; qi < QG; ++qi) {
      if (qi < 2 * NP && qi < (int)v.nq && !(p.debug & 1u)) {
        const float a = acc[qi >> 1][qi & 1];
        const float dist = METRIC == 0 ? a : __fsub_rn(1.0f, a);
        nan_seen |= valid && (dist != dist);
        const uint32_t seq = Src::kSeqIds ? sid : (uint32_t)__builtin_amdgcn_readlane((int)vseq, qi) + row;
        uint64_t cand = valid ? make_key(dist, seq) : kKeyMax;
        if (p.lower != nullptr && cand <= readlane64(vlower, qi)) cand = kKeyMax;  // ranked in an earlier pass
        if (t == 0 && p.k <= 16 && bound[qi] == kKeyMax) wave_topk_fill(list[qi], p.k, cand, lane);
        else wave_topk_update(list[qi], p.k, cand, bound[qi]);
      }
      acc[qi >> 1][qi & 1] = 0.0f;
    }
    if (Src::kSeqIds) {
      const uint32_t nr = (t + 1) * kWave + lane;
      sid = nr < v.nrows ? src.seq_ids(it)[nr] : 0xFFFFFFFFu;
    }
  };

  // Register ring of kBufs chunk buffers: while one is consumed, the loads of the next kBufs-1 chunks
  // are in flight (3 KiB... 24 KiB per wave).  Bytes in flight per CU, not wave count, is what keeps HBM
  // busy while the VALU works: fewer, fatter waves win (measured: 2-deep ring at 3-4 waves/SIMD left
  // memory and math un-overlapped).  The (tile, chunk) walk is flattened into
  // steps and EVERY load is issued unconditionally (steps past the end re-read the last chunk): a
  // branch around a prefetch would make the compiler's vmcnt bookkeeping assume the shorter queue and
  // wait for the prefetch itself.
  // Depth: 4 for the single-query kernels (little per-wave state, memory-bound: 24 KiB in flight per wave);
  // 2 for the batched ones, whose math is bound by the LDS return bus feeding the query operands (one
  // ds_read_b128 per 6 packed instructions, 8 LDS cycles per CU each) and wants 3-4 waves per SIMD more
  // than it wants deeper prefetch (measured both ways).
  constexpr int kBufs = QG == 1 ? 4 : 2;
  u32x4 buf[kBufs][kLoads];
  const uint32_t n_steps = n_tiles * p.n_chunks;
  uint32_t ti = 0, ci = 0;  // (tile, chunk) the next issue fetches
  auto issue_next = [&](u32x4 (&r)[kLoads]) {
    L.template issue<tile_aux<Src::kStreamOnce>()>(r, ti, ci);
    if (ci + 1 < p.n_chunks) ++ci;
    else if (ti + 1 < n_tiles) { ci = 0; ++ti; }  // else: stay on the last chunk (harmless re-read)
  };
  if (n_steps) {
#pragma unroll
    for (int b = 0; b < kBufs - 1; ++b) issue_next(buf[b]);
  }
  uint32_t tc = 0, cc = 0;  // (tile, chunk) being consumed
  for (uint32_t s0 = 0; s0 < n_steps; s0 += kBufs) {
#pragma unroll
    for (int b = 0; b < kBufs; ++b) {
      issue_next(buf[(b + kBufs - 1) % kBufs]);
      if (s0 + b < n_steps) {  // uniform; no vector-memory op inside
        if (!(p.debug & 2u)) tile_chunk_compute<QG, NP, METRIC>(acc, buf[b], v.qb, (p.debug & 4u) ? 0 : cc);
        else acc[0][0] += __uint_as_float(buf[b][0][0] ^ buf[b][kLoads - 1][3]);
        if (++cc == p.n_chunks) {
          cc = 0;
          tile_done(tc++);
        }
      }
    }
  }
#pragma unroll
  for (int qi = 0; qi < QG; ++qi) {
    if (qi < 2 * NP && qi < (int)v.nq) {
      if (lane < (int)p.k) reinterpret_cast<uint64_t*>(readlane64(vout, qi))[lane] = list[qi];
      if (p.bounds != nullptr) {  // publish this item's k-th key (if it has k) as a bound for later items
        const uint64_t kth = readlane64(list[qi], (int)p.k - 1);
        if (kth < bound[qi] && lane == 0) atomicMin((unsigned long long*)(p.bounds + src.bound_slot(it, qi)), (unsigned long long)kth);
      }
    }
  }
}

// One batched work item (QG > 1), two tiles per step.  Same contract as scan_item.
template <int QG, int NP, int METRIC, class Src>
__device__ __forceinline__ void scan_item2(const Src& src, const ScanParams& p, uint32_t it, const ItemView<QG>& v, int lane,
                                           bool& nan_seen) {
  uint64_t list[QG];
#pragma unroll
  for (int qi = 0; qi < QG; ++qi) list[qi] = kKeyMax;
  const uint32_t n_tiles = (v.nrows + kWave - 1) / kWave;
  const uint32_t n_pairs = (n_tiles + 1) / 2;
  TileLoader L;
  L.init(v.rows, (uint64_t)n_tiles * kWave * p.ld * 4u, p.ld, lane);
  f32x2 accA[QG / 2], accB[QG / 2];
#pragma unroll
  for (int p2 = 0; p2 < QG / 2; ++p2) accA[p2] = accB[p2] = f32x2{0.0f, 0.0f};
  uint32_t vseq = 0;
  uint64_t vout = 0;
  uint64_t vbound = kKeyMax;  // shared pruning bound of query `lane`'s merge group (see wave_topk_update)
  uint64_t vlower = 0;        // exclusive lower bound of query `lane` (multi-pass results, see ScanParams::lower)
  uint64_t* bslot = nullptr;
  if (lane < QG && lane < (int)v.nq) {  // per-query constants once, lane qi = query qi (see scan_item)
    vseq = Src::kSeqIds ? 0u : src.seq_base(it, lane);
    vout = (uint64_t)src.out(it, lane);
    if (p.lower != nullptr) vlower = p.lower[src.bound_slot(it, lane)];
    if (p.bounds != nullptr) {
      bslot = p.bounds + src.bound_slot(it, lane);
      vbound = __hip_atomic_load(bslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // stale = prunes less, never wrong
    }
  }
  auto fold = [&](f32x2 (&acc)[QG / 2], uint32_t t) {
    const uint32_t row = t * kWave + lane;
    bool valid = row < v.nrows;
    uint32_t sid = 0;
    if (Src::kSeqIds) {  // (exhaustive scan only: this load sits in the loop and drains the prefetch)
      sid = valid ? src.seq_ids(it)[row] : 0xFFFFFFFFu;
      valid = valid && sid != 0xFFFFFFFFu;
    }
#pragma unroll
    for (int qi = 0; qi < QG; ++qi) {
      if (qi < 2 * NP && qi < (int)v.nq && !(p.debug & 1u)) {
        const float a = acc[qi >> 1][qi & 1];
        const float dist = METRIC == 0 ? a : __fsub_rn(1.0f, a);
        nan_seen |= valid && (dist != dist);
        const uint32_t seq = Src::kSeqIds ? sid : (uint32_t)__builtin_amdgcn_readlane((int)vseq, qi) + row;
        uint64_t cand = valid ? make_key(dist, seq) : kKeyMax;
        if (p.lower != nullptr && cand <= readlane64(vlower, qi)) cand = kKeyMax;  // ranked in an earlier pass
        const uint64_t bnd = p.bounds != nullptr ? readlane64(vbound, qi) : kKeyMax;
        if (t == 0 && p.k <= 16 && bnd == kKeyMax) wave_topk_fill(list[qi], p.k, cand, lane);
        else wave_topk_update(list[qi], p.k, cand, bnd);
      }
      acc[qi >> 1][qi & 1] = 0.0f;
    }
  };
  // 2-deep register ring over (tile pair, chunk) steps, every load unconditional (see scan_item)
  u32x4 buf[2][2 * kLoads];
  const uint32_t last_tile = n_tiles ? n_tiles - 1 : 0;
  const uint32_t n_steps = n_pairs * p.n_chunks;
  uint32_t pi = 0, ci = 0;
  auto issue_next = [&](u32x4 (&r)[2 * kLoads]) {
    const uint32_t tA = 2 * pi, tB = 2 * pi + 1 < n_tiles ? 2 * pi + 1 : last_tile;  // odd tail: B re-reads, rows masked
    const uint32_t soffA = tA * L.tile_bytes + ci * (kLoads * 1024u), soffB = tB * L.tile_bytes + ci * (kLoads * 1024u);
#pragma unroll
    for (int i = 0; i < kLoads; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(L.rsrc, L.lane_off, soffA + (uint32_t)i * 1024u, tile_aux<Src::kStreamOnce>());
#pragma unroll
    for (int i = 0; i < kLoads; ++i) r[kLoads + i] = __builtin_amdgcn_raw_buffer_load_b128(L.rsrc, L.lane_off, soffB + (uint32_t)i * 1024u, tile_aux<Src::kStreamOnce>());
    if (ci + 1 < p.n_chunks) ++ci;
    else if (pi + 1 < n_pairs) { ci = 0; ++pi; }
  };
  if (n_steps) issue_next(buf[0]);
  uint32_t pc = 0, cc = 0;
  const bool stamp = (p.debug & 16u) != 0;  // diagnosis build path: where do a wave's cycles go
  unsigned long long t_wait = 0, t_math = 0, t_fold = 0;
  for (uint32_t s0 = 0; s0 < n_steps; s0 += 2) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      issue_next(buf[b ^ 1]);
      if (s0 + b < n_steps) {  // uniform; no vector-memory op inside (except kSeqIds)
        unsigned long long t0 = 0, t1 = 0, t2 = 0;
        if (stamp) {
          t0 = __builtin_amdgcn_s_memtime();
          __builtin_amdgcn_s_waitcnt(0x4F70);  // vmcnt(16): this step's loads have landed
          t1 = __builtin_amdgcn_s_memtime();
        }
        if (!(p.debug & 2u)) tile_chunk_compute2<QG, NP, METRIC>(accA, accB, buf[b], v.qb, cc);
        else accA[0][0] += __uint_as_float(buf[b][0][0] ^ buf[b][2 * kLoads - 1][3]);
        if (stamp) {
          asm volatile("" :: "v"(accA[0]), "v"(accB[0]));
          t2 = __builtin_amdgcn_s_memtime();
          t_wait += t1 - t0;
          t_math += t2 - t1;
        }
        if (++cc == p.n_chunks) {
          cc = 0;
          fold(accA, 2 * pc);
          fold(accB, 2 * pc + 1);
          ++pc;
          if (stamp) t_fold += __builtin_amdgcn_s_memtime() - t2;
        }
      }
    }
  }
  if (stamp && lane == 0) {
    atomicAdd(p.stamps + 0, t_wait);
    atomicAdd(p.stamps + 1, t_math);
    atomicAdd(p.stamps + 2, t_fold);
    atomicAdd(p.stamps + 4, 1ull);
  }
  uint64_t kth = kKeyMax;  // lane qi collects query qi's k-th key
#pragma unroll
  for (int qi = 0; qi < QG; ++qi)
    if (qi < 2 * NP && qi < (int)v.nq) {
      if (lane < (int)p.k) reinterpret_cast<uint64_t*>(readlane64(vout, qi))[lane] = list[qi];
      const uint64_t kq = readlane64(list[qi], (int)p.k - 1);
      if (lane == qi) kth = kq;
    }
  // publish a full list's k-th key as a bound for the items of the same merge group that start later
  if (bslot != nullptr && kth < vbound) atomicMin((unsigned long long*)bslot, (unsigned long long)kth);
}

// QG == 1: waves are independent (persistent-style stride over items).
// QG  > 1: items come in QUADS that share one query block (same list & query group, four row
// segments; Src pads with empty items): the block's four waves stage the block's interleaved
// query block into LDS once and each scans its own segment against it.
template <int QG, int METRIC, class Src>
__global__ __launch_bounds__(kWave * kWavesPerBlock) void scan_kernel(Src src, ScanParams p) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t n_items = src.n_items();
  bool nan_seen = false;
  const unsigned long long clk0 = (p.debug & 16u) ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rt0 = (p.debug & 16u) ? __builtin_amdgcn_s_memrealtime() : 0ull;
  if constexpr (QG == 1) {
    const uint32_t n_waves = gridDim.x * kWavesPerBlock;
    for (uint32_t it = blockIdx.x * kWavesPerBlock + wid; it < n_items; it += n_waves) {
      ItemView<QG> v;
      src.get(it, v);
      scan_item<1, 1, METRIC>(src, p, it, v, lane, nan_seen);
    }
  } else {
    static_assert(QG == 8 || QG == 16, "query groups are 1, 8 or 16 wide");
    static_assert(kWavesPerBlock == 4, "items are padded to quads");
    extern __shared__ __attribute__((aligned(16))) float qlds[];
    const uint32_t n_quads = n_items / 4;
    const uint32_t n4 = p.ld * (QG / 4);  // float4s of one query block
    // Quads are handed out dynamically (one agent-scope atomic per quad, ~100 us of work each): lists differ
    // 5x in length, a static stride leaves a long tail.  p.next_quad is zeroed by the launcher; nullptr = static.
    uint32_t* nq_lds = reinterpret_cast<uint32_t*>(qlds + (size_t)p.ld * QG);  // one word behind the query block
    for (uint32_t b0 = blockIdx.x;; b0 += gridDim.x) {
      uint32_t bi = b0;
      if (p.next_quad != nullptr) {
        if (threadIdx.x == 0) *nq_lds = atomicAdd(p.next_quad, 1u);
        __syncthreads();
        bi = *nq_lds;  // every wave reads it before the next write: two barriers follow below
      }
      if (bi >= n_quads) break;  // block-uniform
      const uint32_t it = bi * 4 + wid;
      ItemView<QG> v;
      src.get(it, v);  // v.qb / v.nq are the same for the four items of the quad
      const unsigned long long ts0 = (p.debug & 16u) ? __builtin_amdgcn_s_memtime() : 0ull;
      __syncthreads();  // the previous quad's readers are done with the LDS block
      const unsigned long long ts1 = (p.debug & 16u) ? __builtin_amdgcn_s_memtime() : 0ull;
      const f32x4* g = reinterpret_cast<const f32x4*>(v.qb);
      for (uint32_t i = threadIdx.x; i < n4; i += kWave * kWavesPerBlock) reinterpret_cast<f32x4*>(qlds)[i] = g[i];
      __syncthreads();
      if ((p.debug & 16u) && lane == 0) {
        atomicAdd(p.stamps + 3, __builtin_amdgcn_s_memtime() - ts1);  // staging the query block
        atomicAdd(p.stamps + 5, ts1 - ts0);                           // waiting for the quad's slowest wave
        atomicAdd(p.stamps + 6, 1ull);
      }
      v.qb = qlds;
      if (v.nrows == 0) continue;  // padding item (wave-uniform; barriers are outside)
      // Live query pairs: dead pairs are not computed (wave-uniform dispatch), single-pair granularity (odd counts
      // take one ds_read_b64).  option "scan_debug" bit 6 = steps of two pairs, for A/B runs: measured 3.5 % slower
      // on the same box at cfg3.
      const uint32_t np = (p.debug & 64u) ? (((v.nq + 3) >> 2) << 1) : ((v.nq + 1) >> 1);
      if constexpr (QG == 8) {
        switch (np) {
          case 1: scan_item2<8, 1, METRIC>(src, p, it, v, lane, nan_seen); break;
          case 2: scan_item2<8, 2, METRIC>(src, p, it, v, lane, nan_seen); break;
          case 3: scan_item2<8, 3, METRIC>(src, p, it, v, lane, nan_seen); break;
          default: scan_item2<8, 4, METRIC>(src, p, it, v, lane, nan_seen); break;
        }
      } else {
        switch (np) {
          case 1: scan_item2<16, 1, METRIC>(src, p, it, v, lane, nan_seen); break;
          case 2: scan_item2<16, 2, METRIC>(src, p, it, v, lane, nan_seen); break;
          case 3: scan_item2<16, 3, METRIC>(src, p, it, v, lane, nan_seen); break;
          case 4: scan_item2<16, 4, METRIC>(src, p, it, v, lane, nan_seen); break;
          case 5: scan_item2<16, 5, METRIC>(src, p, it, v, lane, nan_seen); break;
          case 6: scan_item2<16, 6, METRIC>(src, p, it, v, lane, nan_seen); break;
          case 7: scan_item2<16, 7, METRIC>(src, p, it, v, lane, nan_seen); break;
          default: scan_item2<16, 8, METRIC>(src, p, it, v, lane, nan_seen); break;
        }
      }
    }
  }
  if constexpr (QG != 1) {
    if ((p.debug & 16u) && blockIdx.x == 0 && threadIdx.x == 0) {  // shader clock = d(memtime)/d(memrealtime) * 100 MHz
      p.stamps[7] = ((__builtin_amdgcn_s_memtime() - clk0) << 20) / ((__builtin_amdgcn_s_memrealtime() - rt0) | 1ull);
    }
  }
  if (__ballot(nan_seen) != 0 && lane == 0) atomicOr(p.status, 1u);
}

// dynamic LDS of a batched scan launch (one interleaved query block) and the resident blocks per CU it allows
inline size_t scan_lds_bytes(int QG, uint32_t ld) { return QG == 1 ? 0 : (size_t)ld * QG * sizeof(float) + 16; }
inline uint32_t scan_blocks_per_cu(int QG, uint32_t ld) {
  if (QG == 1) return 3;                                   // 12 waves/CU at <= 168 VGPRs
  const size_t b = scan_lds_bytes(QG, ld);
  const uint32_t by_lds = (uint32_t)((160u * 1024u) / (b ? b : 1));
  const uint32_t by_vgpr = 2;                              // two tiles per step: <= 256 VGPRs, 8 waves/CU
  return by_lds < 1 ? 1 : (by_lds > by_vgpr ? by_vgpr : by_lds);
}
template <class K>
inline int32_t scan_prepare_launch(K kernel, size_t lds_bytes) {
  if (lds_bytes > 160u * 1024u) return fail(VERS_ERR_INVALID, "vector dimension too large for a batched scan (query block exceeds LDS)");
  if (lds_bytes > 48u * 1024u)
    VERS_HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  return VERS_OK;
}

// ---- merge of partial slots ----------------------------------------------------
// One BLOCK of kMergeWaves waves per output group folds `n_slots` partial slots -- each `k` keys, ASCENDING, kKeyMax
// padded (what scan_item leaves) -- into the group's k smallest keys.  Two facts about sorted slots do the work:
//   * a key of the result is at or below T = the k-th smallest HEAD (the k smallest heads are k keys <= T), so only the
//     slots whose head is <= T -- at most k of them, keys being unique -- can contribute: 1216 slots of a single-query
//     list scan shrink to 10 after one pass over the heads;
//   * two ascending 64-lane lists merge with ONE bitonic merge (reverse one, element-wise minimum, 6 exchange steps)
//     instead of up to 64 ordered inserts.
// (Round 1 offered every key of every slot to a sorted list: plan1_kernel 29 us, ivf_merge_kernel 21 us, flat_merge_kernel
// 23 us of single-block latency around 56-90 us scans.)  Returns the merged list in wave 0, lane i = i-th smallest.
constexpr int kMergeWaves = 16;
__device__ __forceinline__ uint64_t wave_merge2_sorted(uint64_t a, uint64_t b, int lane) {  // both ascending over the lanes
  const uint64_t br = lane_rev64(b, lane);
  uint64_t m = a < br ? a : br;  // bitonic, holds the 64 smallest of the union
  wave_bitonic_merge64(m, lane);
  return m;
}
// A block barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding global access of the wave
// (vmcnt(0)): in the merges below that would stall a wave on loads it has deliberately left in flight (the next stage's
// operands) -- 2 us per barrier at HBM latency.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// cross-wave tree over per-wave ascending lists (through sh); result in wave 0
template <int NW = kMergeWaves>
__device__ __forceinline__ uint64_t block_tree_merge(uint64_t acc, uint64_t (*sh)[kWave], int wid, int lane) {
  lds_barrier();  // `sh` may still be read by a previous call
  sh[wid][lane] = acc;
  lds_barrier();
#pragma unroll
  for (int stride = NW / 2; stride >= 1; stride >>= 1) {
    if (wid < stride) acc = wave_merge2_sorted(acc, sh[wid + stride][lane], lane);
    lds_barrier();
    if (wid < stride) sh[wid][lane] = acc;
    lds_barrier();
  }
  return acc;
}
// NW = waves of the calling block (a power of two; `sh` holds NW rows).  `mid` runs once in every wave after the first round of
// loads has been consumed: the place for a caller's dependent load whose operand was requested before the call.
struct MergeNoOp { __device__ __forceinline__ void operator()() const {} };
// COHERENT: the slots were written by OTHER BLOCKS OF THE SAME LAUNCH (at agent scope: written through); they are read at agent
// scope too -- past whatever this XCD's L2 holds of them -- instead of behind an acquire fence, which would invalidate the L2 and
// send every later load of the block (the plan's tables) to memory as well.
template <int NW = kMergeWaves, class Mid = MergeNoOp, bool COHERENT = false>
__device__ __forceinline__ uint64_t block_merge_keys(const uint64_t* slots_in, uint32_t n_keys, uint32_t k, uint64_t (*sh)[kWave],
                                                     Mid&& mid = Mid()) {
  struct SlotView {
    const uint64_t* p;
    __device__ __forceinline__ uint64_t operator[](uint64_t i) const {
      if constexpr (COHERENT) return __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else return p[i];
    }
  };
  const SlotView slots{slots_in};
  __shared__ uint32_t s_cand[kWave + 2];  // candidate slot ids, [kWave] their count, [kWave + 1] unused
  __shared__ uint64_t s_T;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t n_slots = n_keys / k;
  uint64_t acc = kKeyMax;
  if (n_slots <= (uint32_t)(4 * NW)) {
    // few slots: every wave merges its share (independent loads first), then the tree
    constexpr int U = 4;
    uint64_t v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t sl = (uint32_t)wid + (uint32_t)u * NW;
      v[u] = (sl < n_slots && lane < (int)k) ? slots[(uint64_t)sl * k + lane] : kKeyMax;
    }
    acc = v[0];
#pragma unroll
    for (int u = 1; u < U; ++u) acc = wave_merge2_sorted(acc, v[u], lane);
    mid();
    return block_tree_merge<NW>(acc, sh, wid, lane);
  }
  // (1) + (2): the slots that can contribute are those whose head is at or below T = the k-th smallest head (at most k: keys are
  // unique).  Found in time that does not depend on the data:
  //   (a) every wave loads its share of the heads (all loads before the first use; they stay in registers) and bounds its own
  //       k-th smallest head from above: T0_w = the k-th smallest of its LANE minima (one sort of 64 keys).  T0 = min over the
  //       waves bounds T from above;
  //   (b) the heads at or below T0 -- a few dozen -- go with their slot numbers into ONE list in LDS (an LDS atomic per register
  //       that has any); wave 0 sorts them (one or two sorts of 64), reads T off lane k - 1 and marks the list's entries <= T.
  // More than 128 heads at or below T0 (fewer heads than k per wave, adversarial data): the exact fallback below.
  // (Before: every wave offered its registers one by one to a sorted list -- an ordered insert per head passing the running
  // threshold, 2 us in a wave that meets the best lists' slots first and 6 us in one that meets them last, the block waiting for
  // its slowest wave --, a tree over the waves' lists, and a second pass over the heads against T.)
  __shared__ uint64_t s_lh[128];
  __shared__ uint32_t s_ls[128];
  __shared__ uint64_t s_T0[NW];
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) { s_cand[kWave] = 0u; s_n = 0u; }
  constexpr int kKeep = 16;
  constexpr uint32_t kKept = (uint32_t)kKeep * NW * kWave;
  uint64_t hk[kKeep];
#pragma unroll
  for (int j = 0; j < kKeep; ++j) {
    const uint32_t sl = (uint32_t)j * (NW * kWave) + (uint32_t)wid * kWave + lane;
    hk[j] = sl < n_slots ? slots[(uint64_t)sl * k] : kKeyMax;
  }
  uint64_t vmin = hk[0];
#pragma unroll
  for (int j = 1; j < kKeep; ++j) vmin = hk[j] < vmin ? hk[j] : vmin;
  uint64_t T0w;
  {  // (the k-th smallest LANE MINIMUM: k heads of the wave are at or below it.  A cheaper bound -- the largest of the 16 quad minima,
     // for k <= 16 -- let hundreds of heads through at cfg3, where one wave holds the nearest list's slots and the others do not)
    uint64_t srt = vmin;
    if (n_slots > 128u) wave_bitonic_sort64(srt, lane);     // (block-uniform; at most 128 slots: the list below holds them all)
    T0w = n_slots > 128u ? readlane64(srt, (int)k - 1) : kKeyMax;
  }
  if (lane == 0) s_T0[wid] = T0w;
  mid();
  lds_barrier();
  uint64_t T0 = s_T0[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) T0 = s_T0[w] < T0 ? s_T0[w] : T0;
  auto list_add = [&](uint32_t sl, uint64_t hd) {
    const bool in = hd != kKeyMax && hd <= T0;
    const uint64_t m = __ballot(in);
    if (m) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&s_n, (uint32_t)__popcll(m));
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (in && at < 128u) { s_lh[at] = hd; s_ls[at] = sl; }
    }
  };
#pragma unroll
  for (int j = 0; j < kKeep; ++j)
    if ((uint32_t)j * (NW * kWave) < n_slots) list_add((uint32_t)j * (NW * kWave) + (uint32_t)wid * kWave + lane, hk[j]);  // (block-uniform)
  for (uint32_t s0 = kKept + (uint32_t)wid * kWave; s0 < n_slots; s0 += NW * kWave) {  // (heads beyond the registers: T0 bounds T all the same)
    const uint32_t sl = s0 + lane;
    list_add(sl, sl < n_slots ? slots[(uint64_t)sl * k] : kKeyMax);
  }
  lds_barrier();
  const uint32_t n_list = s_n;
  if (n_list <= 128u) {
    if (wid == 0) {
      const uint64_t h0 = (uint32_t)lane < n_list ? s_lh[lane] : kKeyMax;
      const uint64_t h1 = (uint32_t)lane + kWave < n_list ? s_lh[kWave + lane] : kKeyMax;
      uint64_t srt = h0;
      wave_bitonic_sort64(srt, lane);
      if (n_list > (uint32_t)kWave) {  // (wave-uniform)
        uint64_t s1 = h1;
        wave_bitonic_sort64(s1, lane);
        srt = wave_merge2_sorted(srt, s1, lane);
      }
      const uint64_t T = readlane64(srt, (int)k - 1);  // kKeyMax when fewer than k slots hold a key: every non-empty slot is a candidate
      const bool in0 = h0 != kKeyMax && h0 <= T, in1 = h1 != kKeyMax && h1 <= T;
      const uint64_t m0 = __ballot(in0), m1 = __ballot(in1);
      const uint32_t c0 = (uint32_t)__popcll(m0);
      if (in0) s_cand[(uint32_t)__popcll(m0 & ((1ull << lane) - 1ull))] = s_ls[lane];
      if (in1) s_cand[c0 + (uint32_t)__popcll(m1 & ((1ull << lane) - 1ull))] = s_ls[kWave + lane];
      if (lane == 0) s_cand[kWave] = c0 + (uint32_t)__popcll(m1);
    }
    lds_barrier();
  } else {
    // exact fallback: the waves' k smallest heads by ordered inserts, a tree over the waves, a second pass against T
    uint64_t heads = kKeyMax;
#pragma unroll
    for (int j = 0; j < kKeep; ++j)
      if ((uint32_t)j * (NW * kWave) < n_slots) wave_topk_update(heads, k, hk[j], kKeyMax);  // (block-uniform)
    for (uint32_t s0 = kKept + (uint32_t)wid * kWave; s0 < n_slots; s0 += NW * kWave * 4) {
      uint64_t hd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t sl = s0 + (uint32_t)u * (NW * kWave) + lane;
        hd[u] = sl < n_slots ? slots[(uint64_t)sl * k] : kKeyMax;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) wave_topk_update(heads, k, hd[u], kKeyMax);
    }
    heads = block_tree_merge<NW>(heads, sh, wid, lane);
    if (wid == 0 && lane == (int)k - 1) s_T = heads;
    lds_barrier();
    const uint64_t T = s_T;
    auto offer = [&](uint32_t sl, uint64_t hd) {
      const bool in = hd != kKeyMax && hd <= T;
      const uint64_t m = __ballot(in);
      if (m) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&s_cand[kWave], (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (in) s_cand[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = sl;
      }
    };
#pragma unroll
    for (int j = 0; j < kKeep; ++j) offer((uint32_t)j * (NW * kWave) + (uint32_t)wid * kWave + lane, hk[j]);
    for (uint32_t s0 = kKept + (uint32_t)wid * kWave; s0 < n_slots; s0 += NW * kWave) {
      const uint32_t sl = s0 + lane;
      offer(sl, sl < n_slots ? slots[(uint64_t)sl * k] : kKeyMax);
    }
    lds_barrier();
  }
  const uint32_t n_cand = s_cand[kWave];  // <= k <= 64
  // (3) merge the candidates: wave w takes candidates w, w + NW, ... (four loads in flight at a time)
  for (uint32_t c0 = 0; c0 < n_cand; c0 += 4 * NW) {
    uint64_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t ci = c0 + (uint32_t)wid + (uint32_t)u * NW;
      v[u] = (ci < n_cand && lane < (int)k) ? slots[(uint64_t)s_cand[ci] * k + lane] : kKeyMax;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = wave_merge2_sorted(acc, v[u], lane);
  }
  return block_tree_merge<NW>(acc, sh, wid, lane);
}

}  // namespace vers
