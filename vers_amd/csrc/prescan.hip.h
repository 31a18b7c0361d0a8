// prescan.hip.h -- batched inverted-list scan on the matrix cores with an exact finish (nprobe mode): fp16 shadow rows ->
// v_mfma_f32_32x32x16_f16 by default (half the HBM bytes), f32 rows -> v_mfma_f32_16x16x1_4b_f32 with VERS_SHADOW=0.
//
// The reference's distance is an ordered f32 chain (scan.hip.h); a matrix-core contraction cannot reproduce its
// rounding, so -- exactly as in the coarse quantiser (gemm.hip.h) -- it is used to PRE-SELECT and the result is
// then made exact:
//   (1) prescan_kernel_g: per block a quad of row segments of one
//       list x the <= 32 queries of one group,
//         val[r][n] = |x_r|^2 - 2 <x_r, q_n>          (the row operand exactly as one 16-byte load per lane delivers it)
//       which approximates D_ref(x_r, q_n) - |q_n|^2 within E (below).  Per query the kp = top_k + slack smallest
//       (val, seq) keys are kept in ONE sorted list per block in LDS; the list's
//       last val is the threshold a tile's vals are compared with, shared live with the query's other blocks
//       (atomicMin in HBM, re-read every tile), so that list inserts are rare after the first few tiles.
//   (2) ivf_rescore_kernel: per query, merge the partial lists into the kp globally smallest approximate keys,
//       recompute the distances of the survivors (below) in the reference's own arithmetic (lane per candidate,
//       ordered chain), sort by the exact (distance, seq) key and emit the top_k.  CERTIFICATE: with tau_k the
//       k-th smallest val, a member t of the true top-k has D_t <= (k-th smallest upper bound) <= tau_k + |q|^2 + E,
//       hence val_t <= D_t - |q|^2 + E <= tau_k + 2E; every row outside the kp list has val >= val[kp-1].  So if
//       val[kp-1] > tau_k + 2E (or the list is not full) the true top-k is inside the list -- more precisely among
//       its entries with val <= tau_k + 2E, the survivors -- and the output equals the exact scan bit for bit.
//   (3) fallback_kernel: queries that fail the certificate (ties / near-ties denser
//       than the slack, non-finite values) are queued and re-scanned exactly.  Rare, and never wrong.
//
// E: |val + |q|^2 - D_ref| <= (5 d + 32) u (|q|^2 + max|x|^2), u = 2^-24, d = padded length -- the bound derived
// in gemm.hip.h with one more product in the chain (the |x|^2 term rides through the MFMA as an extra k step) and
// slack for the roundings of the test itself, which is evaluated in f64.  The fp16 shadow adds its MEASURED rounding residual
// (shadow_residual_kernel: max |x - fp16(x)| over the stored rows) and the query split's remainder: ivf_rescore_kernel.
#pragma once
#include <type_traits>

#include "plan.hip.h"
#include "scan.hip.h"
#include "wide.hip.h"

namespace vers {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;

constexpr int kPreQ = 32;        // queries per group: two sets of 16 (one 16x16x1 4-block MFMA covers 64 rows x 16 queries)
// Round 5: the WIDE variant on the fp16 shadow -- 64 queries per block as two sets of 32 (two B operands per row piece), the query
// block as fp16 hi only (64 x 768 x 2 B = 98 KB: the bytes of 32 queries' hi + lo).  A list probed by more than 32 queries of the
// batch was streamed once per group of 32 (streamed / union rows 1.05 at cfg3, 1.5 on an index with one popular list): with 64 per
// group almost every list is streamed ONCE.  Groups of up to 32 run the first set alone (half the MFMAs of the hi + lo block).
constexpr int kPreQWide = 64;
constexpr int kPreCtl = 64;      // entries per control array of a block (cnt | done | thr | locks | pair | sequence base)
constexpr int kPreAux = 2;  // cache policy of the row-tile loads: 2 = nt (streamed once); same-box A/B at cfg3: -1.8 % vs default
constexpr uint32_t kPreMaxKp = 64;  // widest list: one sorted key per lane
// The candidate lists are one key per lane wide: kp = top_k + slack <= 64.  The certificate needs the slack: measured at cfg3 on the fp16 shadow
// (batch 256, scripts/bench_edges.py) top_k <= 48 never failed it; top_k = 58 (slack 6) failed it for 141 of 256 queries -- every one an exact
// re-scan, 18 k q/s where the ordered chains do 64 k.  So results wider than 48 take the ordered chains (rounds 2-5 drew the line at 58).
constexpr uint32_t kPreMinSlack = 16;
constexpr uint32_t kPreMaxP = 1024;  // most probed lists per query on the matrix-core scan (tables and slots grow with b x nprobe)

struct PreParams {
  uint32_t ld, n_chunks, kp;
  uint32_t* status;
  uint32_t* bounds32;   // per merge group: order bits of the smallest known kp-th val (0xFFFFFFFF = none yet)
  uint32_t* qflags;     // per merge group: != 0 -> a non-finite val was seen, the query must be re-done exactly
  uint32_t* next_quad;
  const float* xnorm;   // |x|^2 per storage row
  const uint16_t* rows_bf;  // fp16 shadow of the rows (rows_to_f16_kernel), nullptr: the f32 rows feed the scan
  uint32_t debug;
  uint32_t metric;      // 0: val = |x|^2 - 2<x,q> ~ D_ref - |q|^2 ; 1 (cosine distance 1 - dot): val = -<x,q> ~ D_ref - 1
  unsigned long long* stamps;
};


// f32 tiles -> fp16 shadow tiles (round to nearest even), laid out as the A operand of v_mfma_f32_32x32x16_f16: a
// 64-row tile is ld/16 column blocks x 2 row halves of 1 KiB pieces; in piece (cb, h) lane l holds the 8 columns
// 16*cb + 8*(l >> 5) ... + 7 of row 32*h + (l & 31), so that ONE 16-byte load per lane is the operand of one MFMA
// (32 rows x 16 columns), with no VALU instruction between the load and the matrix core.  Element (r, c) of a tile moves
// from ((c/4)*64 + r)*4 + c%4 (floats) to (((c/16)*2 + r/32)*64 + (c%16)/8*32 + r%32)*8 + c%8 (halves).
// fp16, not bf16: 11 significant bits instead of 8 make the certificate window of the exact finish 8x narrower (round 2's
// first shadow was bf16: ~25 rows per query inside the window at cfg3 and 1 % of the queries beyond the 48 a candidate
// list holds; fp16: as many as with f32 rows).  Elements beyond +-65504 become inf (the row's vals are then non-finite
// and its queries are re-done exactly); tiny ones lose relative precision in fp16's subnormal range -- both are covered
// by shadow_residual_kernel's MEASURED bound rather than a formula.
// Thread = (row, group of 8 columns).
static __global__ void rows_to_f16_kernel(const float* rows, uint32_t ld, uint64_t r_begin, uint64_t r_end, uint16_t* rows_h) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t g8 = ld / 8;
  const uint64_t r = r_begin + t / g8;
  const uint32_t j = (uint32_t)(t % g8);
  if (r >= r_end) return;
  const float* src = rows + (r >> 6) * 64ull * ld;
  f16x8_t out;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(src + ((uint64_t)(2 * j + h) * 64 + (r & 63)) * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) out[4 * h + u] = (_Float16)x[u];  // v_cvt_f16_f32: RNE, inf / NaN preserved
  }
  const uint32_t rr = (uint32_t)(r & 63);
  uint16_t* dst = rows_h + (r >> 6) * 64ull * ld + ((uint64_t)((j >> 1) * 2 + (rr >> 5)) * 64 + (j & 1) * 32 + (rr & 31)) * 8;
  *reinterpret_cast<f16x8_t*>(dst) = out;
}

// max over the stored rows of |x - fp16(x)|^2 (thread per row, like blocked_row_norms_kernel): the exact finish bounds
// the shadow's error of a val by 2 |<x - x~, q>| <= 2 R |q| with this R.  x - fp16(x) is exact in f32 (the two are within
// a factor of two of each other, or the difference is below fp16's subnormal spacing and x itself is tiny); the sum of
// squares is inflated for its own roundings where it is used.  A finite element that overflows fp16 gives R = inf: no
// certificate holds and the shadow switches itself off (vers_ivf::rows_bf).
static __global__ void shadow_residual_kernel(const float* rows, uint32_t ld, const uint32_t* row_ids, uint64_t r_begin, uint64_t r_end,
                                              uint32_t* rmax2_bits) {
  const uint64_t r = r_begin + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= r_end || row_ids[r] == 0xFFFFFFFFu) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(rows + (r >> 6) * 64ull * ld) + (r & 63);
  float acc = 0.0f;
  for (uint32_t j = 0; j < ld / 4; ++j) {
    const f32x4 v = p[(uint64_t)j * 64];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float dlt = v[u] - (float)(_Float16)v[u];
      acc = __fadd_rn(acc, __fmul_rn(dlt, dlt));
    }
  }
  if (acc == acc) atomicMax(rmax2_bits, __float_as_uint(acc));  // acc >= 0: bit order == value order (NaN rows: flagged by their vals)
}

// |x|^2 of every storage row of the blocked matrix (thread per row: consecutive rows are consecutive float4s) and
// the maximum over the rows that hold a vector.
static __global__ void blocked_row_norms_kernel(const float* rows, uint32_t ld, const uint32_t* row_ids, uint64_t r_begin, uint64_t r_end,
                                                float* xnorm, uint32_t* xmax2_bits) {
  const uint64_t r = r_begin + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= r_end) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(rows + (r >> 6) * 64ull * ld) + (r & 63);
  float acc = 0.0f;
  for (uint32_t j = 0; j < ld / 4; ++j) {
    const f32x4 v = p[(uint64_t)j * 64];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __fadd_rn(acc, __fmul_rn(v[u], v[u]));
  }
  // A storage row that holds no vector (slack behind a list, tile padding) is uninitialised memory: its |x|^2 must still be
  // FINITE.  In the fp16 scan |x|^2 enters through v_mfma_f32_32x32x2_f32 with B = 1 on one k-slice and 0 on the other: a real
  // row i gets xn[i] * 1 + xn[32 + i] * 0, and inf * 0 or NaN * 0 of a slack row in the same tile would poison it (seen: an index
  // built into memory a freed index with 1.5e19-sized values had used -- 60-odd of 72 queries non-finite and re-scanned exactly,
  // in one process out of six).  The 4-block f32 MFMA has no such cross term.
  const bool holds_vector = row_ids[r] != 0xFFFFFFFFu;
  xnorm[r] = holds_vector ? acc : 0.0f;
  if (holds_vector && acc == acc) atomicMax(xmax2_bits, __float_as_uint(acc));  // acc >= 0: bit order == value order
}

// ---- the scan kernel: a quad of row segments x one query block per block of eight waves ----------------
// The block owns a quad of row segments of one list and ONE query block in LDS (<= 32 queries as two sets of 16:
// the second set only when the group holds more than 16 queries -- same row operands, so a list probed by up to
// 32 queries is streamed once).  Every segment is walked by a pair of waves (first / second half of its tiles, one
// 64-row tile x 32 columns per step), so each SIMD holds two waves (<= 256 registers each): while one wave sits in
// the back-pressure of an HBM-bound load burst (~100 cycles per load) the other issues its MFMAs -- with one wave
// per SIMD those phases add up (measured 5.80 vs 5.34 ms; that 4-wave variant is in the history of this file).
// The eight waves also share ONE sorted list per query in LDS (a spin lock per query, taken for the ~50
// instructions of an insert): the list sees ~2400 rows instead of an item half's ~220, fills at once, and its last
// val -- the threshold every wave re-reads per tile and the value published to the other blocks of the query -- is
// close to the query's global one.  (With a list per wave the shared threshold stalled near "the kp-th of the best
// 220 rows": ~900 inserts per query and batch instead of the ~170 a perfectly shared threshold needs; inserts
// were 0.38 ms of 5.3 ms.)  One partial slot per (query, list, quad) goes to the exact finish.
constexpr int kPreWavesG = 8;  // 4 items x kPreParts waves (8: two per SIMD, <= 256 registers; 12: three per SIMD, <= 168)
constexpr int kPreWpe = kPreWavesG / 4;  // waves per SIMD the kernel is compiled for (its register budget: 512 / kPreWpe)
constexpr int kPreParts = kPreWavesG / 4;   // waves that share a segment: each walks a contiguous share of its tiles
static_assert(kPreWavesG % 4 == 0 && kPreWavesG >= 4 && kPreWavesG <= 16, "a quad of segments x 1..4 waves each");
// Candidate buffer of a query in LDS: `cap` unsorted keys.  kp <= 40 (top_k <= 30 with the default slack): 64 keys, one
// wave-wide bitonic sort compacts it; wider lists: 128 keys (two sorts + a bitonic merge).  At least 24 free slots after
// every compaction.
// WIDE lists (round 6: kp in (64, kWideMaxKp], results of 49 .. 200 keys): 256 keys, a four-register sort (wide.hip.h).
__host__ __device__ inline uint32_t pre_cap(uint32_t kp) { return kp <= 40u ? 64u : (kp <= 64u ? 128u : 256u); }
// nq = queries per block: kPreQ (32), or 16 -- the NARROW variant for rows too long for a 32-query block (d = 1536: 196 KB
// against the CU's 160 KB of LDS; 16 queries fit up to d = 2304).  Same kernel, same MFMA (half its query columns idle).
// hi_only: the fp16 query block WITHOUT its lo half (2 B per element instead of 4): 32 queries fit up to d = 2304, 16 up to
// d = 4608; the certificate charges the query's measured fp16 residual instead (pre_bound, Rq2).
inline size_t prescan_lds_bytes_g(uint32_t ld, uint32_t kp, uint32_t nq = 32, bool hi_only = false) {  // query block | hand-out word | buffers | cnt, done, thr, locks
  return (size_t)ld * nq * (hi_only ? sizeof(uint16_t) : sizeof(float)) + 16 + (size_t)nq * pre_cap(kp) * sizeof(uint64_t) + 6 * kPreCtl * sizeof(uint32_t);  // (+ pair | sequence base of the quad's queries)
}
constexpr int kPreQNarrow = 16;

// ---- per-query candidate buffers of a block (LDS) ---------------------------------------------------------------
// The eight waves of a block share, per query, an UNSORTED buffer of `cap` keys with a fill counter.  A lane whose
// vals pass the query's threshold reserves slots with ONE LDS atomic add (all 64 lanes, all queries of the set, in a
// single instruction) and stores its keys: no lock, no sorted insert.  Only when a buffer overflows does one wave take
// the query's lock, wait until every reserved slot has been written (`done` counter), sort the buffer across its 64
// lanes (bitonic network on ds_bpermute: ~250 instructions), keep the kp smallest keys and publish the kp-th val as the
// query's new threshold (to this block through LDS, to the query's other blocks through bounds32).  Pruning stays sound
// for the same reason as before: a val above the kp-th smallest of ANY subset of a query's candidates cannot be among
// its kp smallest, and a compaction only drops keys above the kp-th of the buffer.
// (Round 1 kept a sorted list per query and inserted under the lock: 3-5 k cycles per merge, 110-140 us per launch at
// every shard count -- DESIGN.md section 6.  An append is ~10 instructions per candidate.)
// The n_valid (<= cap) keys of a buffer -> ascending over the lanes (kKeyMax padded); lanes >= 64 of a 128-key buffer
// are folded in: the 64 smallest of the union come out.  Whole wave.
__device__ __forceinline__ uint64_t buffer_sorted(const uint64_t* bq, uint32_t n_valid, uint32_t cap, int lane) {
  uint64_t k0 = (uint32_t)lane < n_valid ? bq[lane] : kKeyMax;
  wave_rank_sort64<true>(k0, lane);
  if (cap > (uint32_t)kWave && n_valid > (uint32_t)kWave) {  // (wave-uniform)
    uint64_t k1 = (uint32_t)lane + kWave < n_valid ? bq[kWave + lane] : kKeyMax;
    wave_rank_sort64<true>(k1, lane);
    const uint64_t k1r = lane_rev64(k1, lane);
    k0 = k0 < k1r ? k0 : k1r;  // ascending vs descending: the element-wise minimum holds the 64 smallest, as a bitonic sequence
    wave_bitonic_merge64(k0, lane);
  }
  return k0;
}

// BF: the row operand comes from the fp16 shadow copy (half the HBM bytes) and goes from the load straight into
// v_mfma_f32_32x32x16_f16 (rows_to_f16_kernel writes the shadow in that operand's layout); the query block sits in LDS
// as fp16 hi + lo parts of the scaled f32 query (q' = hi + lo up to 2^-22 |q'|), one MFMA each: the fp16 x fp16 products
// are exact in f32, so val errs by the shadow's own rounding (2^-12 per element) and nothing else of that order.  32 rows
// x 16 columns x 32 queries per 8-pass MFMA: the matrix cores need 1/8 of the f32 path's cycles and the kernel is bound
// by the (halved) HBM stream alone.  The accumulator layout differs from the f32 path's (16x16x1 in 4 blocks: a lane =
// one of 16 query columns x 2 sets): here a lane holds query column lane & 31 and 16 of the 32 rows of a tile half h:
// rows 32*h + 8*(e >> 2) + 4*(lane >> 5) + (e & 3).
// LO (fp16 shadow only): the query block carries the lo half of the hi + lo split (two MFMAs per piece).  LO = false: hi only --
// half the LDS (32 queries per block up to d = 2304: rows that long were scanned by 16-query blocks until round 5, every list
// probed by more than 16 queries streamed again per extra group: 2.15x the union's bytes at d = 1536) and half the MFMAs; what the
// dropped half would have contributed, |<x~, q' - fp16(q')>| <= |x~| |q' - fp16(q')|, is charged to the certificate with the query's
// MEASURED residual (ivf_rescore_kernel sums it next to |q|^2).
// WIDE: candidate lists of more than one key per lane (kp in (64, kWideMaxKp], cap 256): the compaction is a four-register sort.
template <bool BF, int NQ, bool LO, bool WIDE, class Src, class Stage>
__device__ __forceinline__ void prescan_item_g(const Src& src, const PreParams& p, uint32_t it, const ItemView<NQ>& v, int half, int lane,
                                               const float* qm, uint64_t* cbuf, uint32_t* ctl, Stage&& stage) {
  const uint32_t n_tiles = (v.nrows + kWave - 1) / kWave;
  const uint32_t t_per = (n_tiles + kPreParts - 1) / kPreParts;  // (half = the wave's index among those of its segment)
  const uint32_t t_begin = (uint32_t)half * t_per < n_tiles ? (uint32_t)half * t_per : n_tiles;
  const uint32_t t_end = t_begin + t_per < n_tiles ? t_begin + t_per : n_tiles;
  if (t_begin >= t_end) {  // padding item, or a one-tile item's second half (wave-uniform): only the block-wide part
    stage();
    return;
  }
  constexpr int kSets = BF ? (NQ > 32 ? 2 : 1) : 2;   // query columns a lane serves (accumulator layouts: see above)
  constexpr int kSetW = BF ? 32 : 16;                  // query columns per set: one MFMA's N
  constexpr int kAcc = BF ? 2 * kSets : 2;             // accumulators: fp16 rows [set][row half of the tile], f32 rows [set]
  const int n = BF ? (lane & 31) : (lane & 15), quarter = BF ? (lane >> 5) : (lane >> 4);
  const uint32_t kp = p.kp, cap = pre_cap(p.kp);
  uint32_t* const cnt = ctl;                 // [32] slots reserved in the query's buffer (may run past cap: overflow)
  uint32_t* const done = ctl + kPreCtl;        // slots written
  uint32_t* const thrq = ctl + 2 * kPreCtl;    // order bits of the block's threshold of the query (0xFFFFFFFF: none yet)
  uint32_t* const locks = ctl + 3 * kPreCtl;   // compaction locks
  // [32] (query, probe) pair of each query of the quad and [32] the sequence number of that probe's first row: resolved ONCE per
  // staged query block by 32 threads (stage) instead of by every lane of every wave of every item through a chain of five
  // dependent global loads (items -> pair_off -> pairs -> pj_pref, then the bound word): 5-8 us per item, which is what made
  // short items expensive
  const uint32_t* const qpair = ctl + 4 * kPreCtl;
  const uint32_t* const qpref = ctl + 5 * kPreCtl;
  const bool two = kSets == 2 && NQ > kSetW && v.nq > (uint32_t)kSetW;  // wave-uniform: the group reaches into the second set
  const bool stamp = (p.debug & 16u) != 0;
  const unsigned long long tp0 = stamp ? __builtin_amdgcn_s_memtime() : 0ull;
  // the first tile loads go out before anything else: they fly while the item is set up and the block stages
  TileLoader L;
  if (BF) {
    L.init(reinterpret_cast<const float*>(p.rows_bf + (uint64_t)src.storage_row(it) * p.ld), (uint64_t)n_tiles * kWave * p.ld * 2u, p.ld, lane);
    L.tile_bytes = p.ld * 128u;
  } else {
    L.init(v.rows, (uint64_t)n_tiles * kWave * p.ld * 4u, p.ld, lane);
  }
  const uint32_t nch = BF ? p.ld / 64u : p.n_chunks;  // steps per tile: 8 loads = 64 bf16 columns (x 2 row halves) or 32 f32 columns
  const float* xn_item = p.xnorm + src.storage_row(it);
  constexpr int R = 2;  // ring of (tile, chunk) steps, 8 KiB each
  u32x4 buf[R][kLoads];
  float xn[R];
  uint32_t gthr[R][2];
  const uint32_t n_steps = (t_end - t_begin) * nch;
  uint32_t ti = t_begin, ci = 0;
  uint32_t vslot[2] = {0, 0};
  auto issue_next = [&](auto btag, bool with_thr) {
    constexpr int b = decltype(btag)::value;
    if (with_thr) {
      // The query's threshold as its other blocks know it is consumed once per TILE (with the step that completes it); the load
      // stays unconditional -- a branch around a load costs the ring a vmcnt(0) -- but on every other step all lanes read ONE
      // word (slot 0) instead of 32 different ones.  These are agent-scope loads: they bypass the XCD's L2, and 32 separate
      // lines per step and wave were ~400 memory-side transactions per 96 KB tile next to the tile's own 96 (round 3 loaded
      // them with every step).
      const bool fin = ci + 1 >= nch;
      gthr[b][0] = __hip_atomic_load(p.bounds32 + (fin ? vslot[0] : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      gthr[b][1] = kSets == 2 ? __hip_atomic_load(p.bounds32 + (fin ? vslot[1] : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
    } else {
      gthr[b][0] = gthr[b][1] = 0xFFFFFFFFu;
    }
    xn[b] = xn_item[ti * kWave + lane];
    const uint32_t soff = ti * L.tile_bytes + ci * (kLoads * 1024u);
#pragma unroll
    for (int i = 0; i < kLoads; ++i) buf[b][i] = __builtin_amdgcn_raw_buffer_load_b128(L.rsrc, L.lane_off, soff + (uint32_t)i * 1024u, kPreAux);
    if (ci + 1 < nch) ++ci;
    else if (ti + 1 < t_end) { ci = 0; ++ti; }
  };
  issue_next(std::integral_constant<int, 0>{}, false);

  bool live[2];
  uint32_t vseq[2] = {0, 0};
  float thr[2];
  f32x16_t acc[kAcc];
  bool bad = false;
  // The eight waves of the block walk their tiles at the same pace and meet the same full buffers: every wave starts
  // its round over the overflowed query columns at its own offset, so that they do not all queue for the same lock.
  const int rot = (int)(((it & 3u) * (uint32_t)kPreParts + (uint32_t)half) << 1);

  // End of a tile for query set S: acc already holds val (the |x|^2 term went through the matrix core).  Each lane holds
  // 16 vals of ONE query column (lane & 15), rows 16*(e>>2) + 4*quarter + (e&3): one compare per register against the
  // lane's threshold decides whether anything happens at all.
  // (BF: S = 0 and `h` is the row half of the tile the accumulator covers)
  auto fold = [&](auto set_tag, f32x16_t& a, uint32_t t, uint32_t h) {
    constexpr int S = decltype(set_tag)::value;
    constexpr uint32_t kRowStep = BF ? 8u : 16u;  // rows between the lane's register groups of four
    const uint32_t r0 = t * kWave + 32u * h + 4u * (uint32_t)quarter;
    if ((t + 1) * kWave > v.nrows) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const uint32_t row = r0 + kRowStep * (e >> 2) + (e & 3);
        if (row >= v.nrows) a[e] = __builtin_nanf("");
        else bad |= live[S] && !(__builtin_fabsf(a[e]) < __builtin_inff());
      }
    } else if (live[S]) {
#pragma unroll
      for (int e = 0; e < 16; ++e) bad |= !(__builtin_fabsf(a[e]) < __builtin_inff());
    }
    // Which of the lane's 16 vals pass its query's threshold: one bit per register.  (A query without a threshold yet
    // -- thr = +inf -- passes all its finite vals; the first overflow's compaction gives it one.)
    uint32_t pm = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) pm |= a[e] <= thr[S] ? 1u << e : 0u;
    if (__ballot(pm != 0) != 0 && !(p.debug & 1u)) {
      const uint32_t q = (uint32_t)(S * kSetW + n);
      uint64_t* const bq = cbuf + (size_t)q * cap;
      const uint32_t sq0 = vseq[S] + r0;
      // reserve + store: every lane for its own query column, the whole wave in one LDS atomic.  Bits of `pend` that found
      // a slot are cleared; the others (buffer full) stay for the slow path.
      auto append = [&](uint32_t& pend) {
        const uint32_t c = (uint32_t)__popc(pend);
        uint32_t pos = 0, nw = 0;
        if (c) pos = __hip_atomic_fetch_add(cnt + q, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if (pend >> e & 1u) {
            if (pos < cap) {
              bq[pos] = make_key(a[e], sq0 + kRowStep * (e >> 2) + (e & 3));
              pend &= ~(1u << e);
              ++nw;
            }
            ++pos;
          }
        }
        if (nw) __hip_atomic_fetch_add(done + q, nw, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);  // (after the stores)
      };
      uint32_t pend = pm;
      append(pend);
      if (stamp) {
        const unsigned long long nc = (unsigned long long)__popcll(__ballot(pm != 0));
        if (lane == 0) atomicAdd(p.stamps + 13, nc);
      }
      uint64_t ovf = (p.debug & 4096u) ? 0ull : __ballot(pend != 0);  // (diagnosis, option "scan_debug" & 4096: candidates that find their buffer full are DROPPED -- wrong results, the appends' cost without the compactions')
      // Some query's buffer is full.  ONE wave compacts it (the lock decides which); every other wave with candidates for that
      // query only waits for the counter to re-open and then places what is still worth placing WITHOUT the lock -- appends
      // never needed it.  (Round 3 made every such wave take the lock in turn, compaction or not: the eight waves of a block
      // meet the same full buffers at the same time, and at 8 ranks -- looser thresholds, 17 k compactions and 300 k candidates
      // per launch -- that queueing was most of the 55 us the list inserts cost a 370 us launch.)
      int rot_cur = rot;  // (wave-uniform) where this wave's round over the overflowed query columns starts
      while (ovf) {
        const uint64_t ovr = rot_cur ? (ovf >> rot_cur) | (ovf << (64 - rot_cur)) : ovf;
        const int L = (__ffsll((unsigned long long)ovr) - 1 + rot_cur) & 63;
        const uint32_t qq = (uint32_t)__builtin_amdgcn_readlane((int)q, L);
        const bool mine = q == qq && pend != 0;
        uint64_t* const bqq = cbuf + (size_t)qq * cap;
        uint32_t cv = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(cnt + qq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (cv >= cap) {
          uint32_t got = 0;
          if (lane == 0) got = __hip_atomic_exchange(locks + qq, 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u ? 1u : 0u;
          got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
          if (!got) {
            // Somebody else is compacting this buffer (or was, a moment ago).  Rather than wait here, go on with the NEXT query
            // column that has something pending (its buffer may be free or need a compactor: useful work either way) and come
            // back to this one on the way round; only when nothing else is pending does the wave sleep before it looks again --
            // at the counter AND, if the buffer is still full, at the lock.
            const int next = (L + 1) & 63;
            const uint64_t others = __ballot(pend != 0 && q != qq);  // (the lanes of the OTHER query columns with something pending)
            if (others == 0) __builtin_amdgcn_s_sleep(1);
            rot_cur = next;
            continue;
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          cv = (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(cnt + qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
          if (cv >= cap) {  // (still full under the lock: this wave compacts)
            // every reservation below cap belongs to a wave that is on its way to store it without needing this lock
            while ((uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(done + qq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) != cap)
              __builtin_amdgcn_s_sleep(1);
            uint32_t kb;
            if constexpr (WIDE) {
              // 256 keys, four per lane: sorted across the registers (wide.hip.h), the kp smallest go back in order
              uint64_t wk[kWideR];
#pragma unroll
              for (int r = 0; r < kWideR; ++r) wk[r] = bqq[r * kWave + lane];
              wide_sort<true>(wk, lane);
#pragma unroll
              for (int r = 0; r < kWideR; ++r)
                if ((uint32_t)(r * kWave + lane) < kp) bqq[r * kWave + lane] = wk[r];
              kb = (uint32_t)(wide_get(wk, kp - 1u) >> 32);  // cap keys >= kp: always a real key
            } else if (cap == (uint32_t)kWave) {
              // A full 64-key buffer, one key per lane: the kp smallest by RANK COUNTING -- rank = how many of the 64 keys are
              // smaller (keys are unique: (val, seq)), 64 x (two v_readlane, one 64-bit compare, one add), no LDS round trips --
              // instead of a bitonic sort through ds_bpermute (21 dependent stages of two permutes: ~3x the cycles, all of them
              // under the query's lock with the query's other waves waiting; at 8 ranks a launch makes 17 k of these).  The key of
              // rank r goes to slot r, so the kept prefix comes out sorted like before.
              const uint64_t mykey = bqq[lane];
              uint32_t rank = 0;
#pragma unroll
              for (int j = 0; j < kWave; ++j) rank += readlane64(mykey, j) < mykey ? 1u : 0u;
              if (rank < kp) bqq[rank] = mykey;
              const uint64_t at = __ballot(rank == kp - 1u);  // (exactly one lane)
              kb = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mykey >> 32), __ffsll((unsigned long long)at) - 1);
            } else {
              const uint64_t srt = buffer_sorted(bqq, cap, cap, lane);
              if (lane < (int)kp) bqq[lane] = srt;
              kb = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(srt >> 32), (int)kp - 1);  // cap keys >= kp: always a real key
            }
            if (lane == 0) {
              __hip_atomic_store(thrq + qq, kb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_store(done + qq, kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // buffer, done and threshold before the counter re-opens it
            if (lane == 0) __hip_atomic_store(cnt + qq, kp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane == (int)(BF ? (qq & 31u) : (qq & 15u)) && !(p.debug & 8192u)) atomicMin(p.bounds32 + vslot[S], kb);
            if (stamp && lane == 0) atomicAdd(p.stamps + 12, 1ull);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) __hip_atomic_store(locks + qq, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        uint32_t mp = 0;
        if (mine) {  // the threshold moved: most of what was pending is no longer a candidate
          const uint32_t bh = __hip_atomic_load(thrq + qq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (bh != 0xFFFFFFFFu) {
            const float g = __uint_as_float(order_bits_to_f32_bits(bh));
            thr[S] = g < thr[S] ? g : thr[S];
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) mp |= ((pend >> e & 1u) && a[e] <= thr[S]) ? 1u << e : 0u;
        }
        append(mp);
        if (mine) pend = mp;
        ovf = __ballot(pend != 0);
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = 0.0f;
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;

  stage();
  // (the accumulators start behind the block-wide part: across it they would be 32 live registers at the kernel's register peak)
#pragma unroll
  for (int e = 0; e < 16; ++e)
#pragma unroll
    for (int a_ = 0; a_ < kAcc; ++a_) acc[a_][e] = 0.0f;
  // (after the block-wide part: the quad's pair / sequence-base table in LDS is what stage() filled -- or left, for the next quad of a run)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    live[s] = s < kSets && s * kSetW + n < (int)v.nq && s * kSetW + n < NQ;  // (narrow blocks: the MFMA's query columns 16 .. 31 repeat 0 .. 15 and are ignored)
    thr[s] = -__builtin_inff();  // dead query columns never hit
    if (live[s]) {
      vseq[s] = qpref[s * kSetW + n] + v.row0;
      vslot[s] = src.slot_of_pair(qpair[s * kSetW + n]);
      const uint32_t b0 = __hip_atomic_load(p.bounds32 + vslot[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      thr[s] = b0 == 0xFFFFFFFFu ? __builtin_inff() : __uint_as_float(order_bits_to_f32_bits(b0));
    }
  }
  uint32_t tc = t_begin, cc = 0;
  unsigned long long t_math = 0, t_fold = 0, t_issue = 0;
  const unsigned long long tp1 = stamp ? __builtin_amdgcn_s_memtime() : 0ull;
  uint32_t fold_tile = 0;
  auto run_folds = [&]() {
    if constexpr (BF) {
      fold(Set0{}, acc[0], fold_tile, 0u);
      fold(Set0{}, acc[1], fold_tile, 1u);
      if constexpr (kSets == 2)
        if (two) {
          fold(Set1{}, acc[kAcc - 2], fold_tile, 0u);
          fold(Set1{}, acc[kAcc - 1], fold_tile, 1u);
        }
    } else {
      fold(Set0{}, acc[0], fold_tile, 0u);
      if (two) fold(Set1{}, acc[1], fold_tile, 0u);
    }
  };
  auto step = [&](auto btag, uint32_t s0) {
    constexpr int B = decltype(btag)::value;
    const unsigned long long ti0 = stamp ? __builtin_amdgcn_s_memtime() : 0ull;
    issue_next(std::integral_constant<int, (B + R - 1) % R>{}, true);
    if (s0 + B < n_steps) {
      unsigned long long t1 = 0, t2 = 0;
      if (stamp) { t1 = __builtin_amdgcn_s_memtime(); t_issue += t1 - ti0; }
      if (!(p.debug & 2u)) {
        if constexpr (!BF) {
          const f32x4* ql = reinterpret_cast<const f32x4*>(qm) + ((size_t)cc * kLoads * NQ + n);
#pragma unroll
          for (int i = 0; i < kLoads; ++i) {
            const f32x4 q4 = ql[i * NQ];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[0] = __builtin_amdgcn_mfma_f32_16x16x1f32(__uint_as_float(buf[B][i][u]), q4[u], acc[0], 0, 0, 0);
          }
          if (two) {
#pragma unroll
            for (int i = 0; i < kLoads; ++i) {
              const f32x4 q4 = ql[i * NQ + 16];
#pragma unroll
              for (int u = 0; u < 4; ++u) acc[1] = __builtin_amdgcn_mfma_f32_16x16x1f32(__uint_as_float(buf[B][i][u]), q4[u], acc[1], 0, 0, 0);
            }
          }
        } else {
          // query block: [hi | lo][column block of 16][64 lanes] x 16 bytes, lane = octet * 32 + query (the B operand's layout)
          // (narrow blocks: 16 query slots per octet; lane (octet, n) reads slot n & 15)
          constexpr int kCb = 2 * NQ;  // f16x8 entries per column block: 2 octets x NQ query slots
          const f16x8_t* qh = reinterpret_cast<const f16x8_t*>(qm) + ((size_t)cc * (kLoads / 2) * kCb + (lane >> 5) * NQ + (n & (NQ - 1)));
          const f16x8_t* ql = qh + (size_t)(p.ld / 16u) * kCb;
#pragma unroll
          for (int cb = 0; cb < kLoads / 2; ++cb) {
            const f16x8_t bh = qh[cb * kCb];
            f16x8_t bl = bh;
            if constexpr (LO) bl = ql[cb * kCb];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
              const f16x8_t ar = __builtin_bit_cast(f16x8_t, buf[B][2 * cb + hf]);
              if constexpr (LO) acc[hf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar, bl, acc[hf], 0, 0, 0);  // (small term first)
              acc[hf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar, bh, acc[hf], 0, 0, 0);
            }
            if constexpr (kSets == 2)
              if (two) {  // the second set of 32 queries: the same row pieces against their B operand (32 slots further)
                const f16x8_t bh2 = qh[cb * kCb + 32];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                  acc[kAcc - 2 + hf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, buf[B][2 * cb + hf]), bh2, acc[kAcc - 2 + hf], 0, 0, 0);
              }
          }
        }
      } else {
        acc[0][0] += __uint_as_float(buf[B][0][0] ^ buf[B][kLoads - 1][3]);
      }
      if (stamp) {
        asm volatile("" ::"v"(acc[0][0]));
        t2 = __builtin_amdgcn_s_memtime();
        t_math += t2 - t1;
      }
      if (++cc == nch) {
        cc = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (live[s]) {
            if (gthr[B][s] != 0xFFFFFFFFu) {  // other blocks of the query
              const float g = __uint_as_float(order_bits_to_f32_bits(gthr[B][s]));
              thr[s] = g < thr[s] ? g : thr[s];
            }
            // this block's threshold of the query (set by whichever wave compacted its buffer last)
            const uint32_t bh = __hip_atomic_load(thrq + (uint32_t)(s * kSetW + n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (bh != 0xFFFFFFFFu) {
              const float g = __uint_as_float(order_bits_to_f32_bits(bh));
              thr[s] = g < thr[s] ? g : thr[s];
            }
          }
        }
        const float nrm = p.metric ? 0.0f : 1.0f;  // (the cosine-distance val has no |x|^2 term; 0 * NaN still flags a bad row)
        if constexpr (BF) {
          // + |x_row|^2 for every query column: A[i][k] = xn of lane (i, k) = row 32k + i of the tile, B[k][j] = (k == h)
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xn[B], quarter == 0 ? nrm : 0.0f, acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xn[B], quarter == 1 ? nrm : 0.0f, acc[1], 0, 0, 0);
          if constexpr (kSets == 2)
            if (two) {
              acc[kAcc - 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(xn[B], quarter == 0 ? nrm : 0.0f, acc[kAcc - 2], 0, 0, 0);
              acc[kAcc - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xn[B], quarter == 1 ? nrm : 0.0f, acc[kAcc - 1], 0, 0, 0);
            }
        } else {
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x1f32(xn[B], nrm, acc[0], 0, 0, 0);  // + |x_row|^2 for every query column
          if (two) acc[1] = __builtin_amdgcn_mfma_f32_16x16x1f32(xn[B], nrm, acc[1], 0, 0, 0);
        }
        fold_tile = tc;
        ++tc;
        run_folds();
        if (stamp) t_fold += __builtin_amdgcn_s_memtime() - t2;
      }
    }
  };
  for (uint32_t s0 = 0; s0 < n_steps; s0 += R) {
    step(std::integral_constant<int, 0>{}, s0);
    step(std::integral_constant<int, 1>{}, s0);
  }
  const unsigned long long te0 = stamp ? __builtin_amdgcn_s_memtime() : 0ull;
  if (stamp && lane == 0) {
    atomicAdd(p.stamps + 1, t_math);
    atomicAdd(p.stamps + 2, t_fold);
    atomicAdd(p.stamps + 4, 1ull);
    atomicAdd(p.stamps + 8, t_issue);
    atomicAdd(p.stamps + 9, tp1 - tp0);
    atomicAdd(p.stamps + 10, te0 - tp1);
  }
  if (bad) {
    if (live[0]) p.qflags[vslot[0]] = 1u;
    if (live[1]) p.qflags[vslot[1]] = 1u;
  }
  if (stamp && lane == 0) atomicAdd(p.stamps + 11, __builtin_amdgcn_s_memtime() - te0);
}

template <bool BF, int NQ, class Src, bool LO = true, bool WIDE = false>
__global__ __launch_bounds__(kWave * kPreWavesG) __attribute__((amdgpu_waves_per_eu(kPreWpe, kPreWpe))) void prescan_kernel_g(Src src, PreParams p) {
  static_assert(NQ == kPreQ || NQ == kPreQNarrow || (NQ == kPreQWide && BF && !LO), "32 queries per block, the narrow variant's 16, or 64 with the hi-only block on the shadow");
  static_assert(BF || LO, "the hi-only query block belongs to the fp16 shadow");
  static_assert(!WIDE || (BF && !LO && NQ <= kPreQ), "wide lists: fp16 shadow, hi-only query blocks of 32 or 16 queries (256 keys x 32 queries = 64 KB of buffers)");
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  extern __shared__ __attribute__((aligned(16))) float qlds[];
  const size_t q_floats = LO ? (size_t)p.ld * NQ : (size_t)p.ld * NQ / 2;  // the query block: f32 / fp16 hi + lo (4 B per element), or fp16 hi only
  uint32_t* nq_lds = reinterpret_cast<uint32_t*>(qlds + q_floats);
  const uint32_t cap = pre_cap(p.kp);
  uint64_t* buf = reinterpret_cast<uint64_t*>(qlds + q_floats + 4);  // [NQ queries][cap] candidate keys, shared by the 8 waves
  uint32_t* ctl = reinterpret_cast<uint32_t*>(buf + (size_t)NQ * cap);         // cnt | done | thr | locks, [32] each
  const uint32_t n_quads = src.n_items() / 4;
  const unsigned long long clk0 = (p.debug & 16u) ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long rt0 = (p.debug & 16u) ? __builtin_amdgcn_s_memrealtime() : 0ull;
  // Quads are handed out in RUNS (guided self-scheduling): a block takes `remaining / (2 * blocks)` consecutive quads at a time,
  // at most kMaxRun, down to one at the end of the launch -- long stretches while there is plenty of work, single quads when
  // the last CUs are being filled.  Consecutive quads of one (list, query group) share their query block and their candidate
  // buffers: the block stages ONCE for them and leaves ONE partial slot (the first quad's; the others' are written empty).
  // That is what lets the host cut the lists of a SHARDED scan into finer quads (2.7 whole-list quads per CU at 8 ranks left
  // the last third of the launch half empty) without paying the 6 us of staging per quad.
  constexpr uint32_t kMaxRun = 8;
  uint32_t run_first = 0xFFFFFFFFu, run_last = 0, prev_nq = 0;  // the quads whose lists still sit in LDS (one merged run)
  uint32_t cur_list = 0xFFFFFFFFu, cur_group = 0;
  auto write_out = [&]() {  // block-wide, between barriers: the finished run's buffers -> sorted kp keys in its first quad's partial slots
    if (run_first == 0xFFFFFFFFu) return;
    for (uint32_t qi = (uint32_t)wid; qi < prev_nq; qi += kPreWavesG) {  // a wave per query
      const uint32_t cv = ctl[qi];
      if constexpr (WIDE) {  // up to 256 keys -> ascending over four registers; slot entry e = register e / 64 of lane e % 64
        const uint32_t nv = cv < cap ? cv : cap;
        const uint64_t* bq = buf + (size_t)qi * cap;
        uint64_t wk[kWideR];
#pragma unroll
        for (int r = 0; r < kWideR; ++r) wk[r] = (uint32_t)(r * kWave + lane) < nv ? bq[r * kWave + lane] : kKeyMax;
        wide_sort<true>(wk, lane);
#pragma unroll
        for (int r = 0; r < kWideR; ++r) {
          const uint32_t e = (uint32_t)(r * kWave + lane);
          if (e < p.kp) {
            src.out_quad(run_first * 4, (int)qi)[e] = wk[r];
            for (uint32_t b = run_first + 1; b <= run_last; ++b) src.out_quad(b * 4, (int)qi)[e] = kKeyMax;
          }
        }
        const uint64_t lastk = wide_get(wk, p.kp - 1u);
        if (lane == 0 && lastk != kKeyMax && !(p.debug & 8192u)) atomicMin(p.bounds32 + src.bound_slot(run_first * 4, (int)qi), (uint32_t)(lastk >> 32));
        continue;
      }
      const uint64_t srt = buffer_sorted(buf + (size_t)qi * cap, cv < cap ? cv : cap, cap, lane);
      if (lane < (int)p.kp) {
        src.out_quad(run_first * 4, (int)qi)[lane] = srt;
        for (uint32_t b = run_first + 1; b <= run_last; ++b) src.out_quad(b * 4, (int)qi)[lane] = kKeyMax;  // (the exact finish reads every slot of a scanned list)
      }
      if (lane == (int)p.kp - 1 && srt != kKeyMax && !(p.debug & 8192u))  // a full list: its last val bounds the query's kp-th smallest
        atomicMin(p.bounds32 + src.bound_slot(run_first * 4, (int)qi), (uint32_t)(srt >> 32));
    }
  };
  const uint32_t n_res = gridDim.x;
  for (uint32_t b0 = blockIdx.x;; b0 += gridDim.x) {
    uint32_t start = b0, count = 1;
    if (p.next_quad != nullptr) {
      if (threadIdx.x == 0) {
        const uint32_t seen = __hip_atomic_load(p.next_quad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t r = seen < n_quads ? (n_quads - seen) / (2u * n_res) : 1u;
        r = r < 1u ? 1u : (r > kMaxRun ? kMaxRun : r);
        nq_lds[0] = atomicAdd(p.next_quad, r);
        nq_lds[1] = r;
      }
      __syncthreads();
      start = nq_lds[0]; count = nq_lds[1];
    }
    if (start >= n_quads) break;
    const uint32_t end = start + count < n_quads ? start + count : n_quads;
    for (uint32_t bi = start; bi < end; ++bi) {
    const ItemDesc d0 = src.items[4 * bi];
    // (block-uniform.  Only the quad right behind the run's last one: the slots written empty must be this block's own)
    const bool cont = run_first != 0xFFFFFFFFu && bi == run_last + 1 && d0.list == cur_list && d0.group == cur_group;
    const uint32_t it = bi * 4 + (wid & 3);
    ItemView<NQ> v;
    src.get(it, v);
    const float qscale = p.metric ? -1.0f : -2.0f;
    // The quad's query block: <= 32 padded queries gathered from their rows, scaled by -2 (-1: cosine distance), in the MFMA operand
    // layout l4[column group * 32 + slot].  Every thread serves ONE slot (512 % 32 == 0; 16 slots when the group
    // holds <= 16 queries, so that all threads load) and its row pointer is resolved here, ahead of the barrier.
    const uint32_t ns = (NQ > 32 && v.nq > 32u) ? 64u : ((NQ > 16 && (BF || v.nq > 16)) ? 32u : 16u);
    const uint32_t slot = threadIdx.x & (ns - 1u), cg0 = threadIdx.x / ns, cg_step = (kWave * kPreWavesG) / ns;
    const float* qrow = (!cont && slot < v.nq) ? src.query_row(it, slot) : nullptr;
    auto stage = [&]() {
      const unsigned long long ts0 = (p.debug & 16u) ? __builtin_amdgcn_s_memtime() : 0ull;
      // the first kStageU loads of every thread go out BEFORE the barriers (they fly while the slowest wave of the
      // previous quad finishes and its lists are written out); one load at a time behind the barriers was ~7 us per
      // quad at d = 768 -- 7 % of the launch
      constexpr int kStageU = (WIDE ? 16 : 96) / kPreWavesG;  // (default: x the block's threads / 32 slots = 192 column groups: d <= 768 in one round; the WIDE kernel has no registers to spare: two rounds)
      const uint32_t n_cg = p.ld / 4u;
      f32x4 x[kStageU];
#pragma unroll
      for (int u = 0; u < kStageU; ++u) {
        const uint32_t cg = cg0 + (uint32_t)u * cg_step;
        x[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (qrow != nullptr && cg < n_cg) x[u] = *reinterpret_cast<const f32x4*>(qrow + 4 * cg);
      }
      __syncthreads();  // every wave is done with the previous quad: its lists are final
      const unsigned long long ts1 = (p.debug & 16u) ? __builtin_amdgcn_s_memtime() : 0ull;
      write_out();
      __syncthreads();
      if (threadIdx.x < 4 * kPreCtl) ctl[threadIdx.x] = (threadIdx.x / kPreCtl) == 2 ? 0xFFFFFFFFu : 0u;  // empty buffers, no threshold, locks open
      if (threadIdx.x < v.nq) {  // the quad's queries: their (query, probe) pair and the sequence number of the probe's first row
        const uint32_t pr = src.pair_of(bi * 4, (int)threadIdx.x);
        ctl[4 * kPreCtl + threadIdx.x] = pr;
        ctl[5 * kPreCtl + threadIdx.x] = src.pj_pref[pr];
      }
      f32x4* l4 = reinterpret_cast<f32x4*>(qlds);
      auto put = [&](uint32_t cg, const f32x4& y) {  // columns 4*cg .. 4*cg + 3 of the thread's query, already scaled
        if constexpr (BF) {
          // hi = fp16(y), lo = fp16(y - hi) into the B operand layout of the 32x32x16 MFMA (prescan_item_g)
          f16x4_t hi, lo;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            hi[u] = (_Float16)y[u];
            lo[u] = (_Float16)(y[u] - (float)hi[u]);
          }
          _Float16* qb = reinterpret_cast<_Float16*>(qlds);
          const uint32_t at = (((cg >> 2) * 2u + ((cg >> 1) & 1u)) * (uint32_t)NQ + slot) * 8u + (cg & 1u) * 4u;
          *reinterpret_cast<f16x4_t*>(qb + at) = hi;
          if constexpr (LO) *reinterpret_cast<f16x4_t*>(qb + (size_t)p.ld * NQ + at) = lo;
        } else {
          l4[cg * NQ + slot] = y;
        }
      };
#pragma unroll
      for (int u = 0; u < kStageU; ++u) {
        const uint32_t cg = cg0 + (uint32_t)u * cg_step;
        if (cg < n_cg) put(cg, qscale * x[u]);
      }
      for (uint32_t cg = cg0 + kStageU * cg_step; cg < n_cg; cg += cg_step) {  // (wider rows than kStageU rounds cover)
        f32x4 y = {0.0f, 0.0f, 0.0f, 0.0f};
        if (qrow != nullptr) y = *reinterpret_cast<const f32x4*>(qrow + 4 * cg);
        put(cg, qscale * y);
      }
      __syncthreads();
      if ((p.debug & 16u) && lane == 0) {
        atomicAdd(p.stamps + 3, __builtin_amdgcn_s_memtime() - ts1);
        atomicAdd(p.stamps + 5, ts1 - ts0);
        atomicAdd(p.stamps + 6, 1ull);
      }
    };
    if (cont) {  // same query block, same buffers: nothing to stage, nothing to wait for
      prescan_item_g<BF, NQ, LO, WIDE>(src, p, it, v, wid >> 2, lane, qlds, buf, ctl, [] {});
      run_last = bi;
    } else {
      prescan_item_g<BF, NQ, LO, WIDE>(src, p, it, v, wid >> 2, lane, qlds, buf, ctl, stage);
      run_first = run_last = bi;
      cur_list = d0.list; cur_group = d0.group;
      prev_nq = v.nq;
    }
    }  // quads of the run
  }
  __syncthreads();
  write_out();
  if ((p.debug & 16u) && blockIdx.x == 0 && threadIdx.x == 0)
    p.stamps[7] = ((__builtin_amdgcn_s_memtime() - clk0) << 20) / ((__builtin_amdgcn_s_memrealtime() - rt0) | 1ull);
}

// ---- the certificate's bound (host + device: the kernel evaluates it per candidate, vers_ivf_test_last_vals per dumped val) ---
// How far a pre-filter value `val` (|x|^2 - 2 <x~, q> on the matrix cores, or -<x~, q> for the cosine distance) can be from
// the reference's ordered-chain distance of the same row: | val + |q|^2 - D_ref |  (cosine: | 1 + val - D_ref |), u = 2^-24.
//   (1) the reference's own chain: sub, mul and d adds, each rounded: |D_ref - T| <= (d + 2) u T / (1 - (d + 2) u), T = the exact
//       squared distance of the f32 operands -- PER CANDIDATE, T <= val + |q|^2 + (2..5) (round 2 charged every candidate the
//       global 2 (d + 2) u (|q|^2 + max |x|^2): a row at distance 0.6 paid for one at 4);
//   (2) |x|^2 as stored (xnorm, an ordered f32 sum) and |q|^2 as the finish sums it (any order): (d + 1) u each, of max |x|^2
//       and |q|^2;
//   (3) the matrix cores: products of two 16-bit (or f32 x f32 into f32) operands accumulated in f32 -- modelled as d
//       roundings of u sum |x_i q'_i| <= u |x||q'| (Cauchy-Schwarz), q' = -2q.  MEASURED on this chip: at most 9 (f16, bf16)
//       / 25 (f32) such roundings over d = 768 for adversarial operands, fp16 subnormals are not flushed
//       (tests/test_mfma_model_gpu.py, profiles/r03_mfma_model.txt) -- a 30x margin under the model's d;
//   (4) |x|^2 riding through the matrix core as one more k-step, the final f32 of the accumulator: a few u (|q|^2 + max|x|^2);
//   (5) fp16 shadow rows: 2 |<x - x~, q>| <= 2 R |q| with R = the MEASURED largest |x - fp16(x)| over the stored rows
//       (shadow_residual_kernel), and the query's fp16 hi + lo split leaving <= 2^-22 |q'_j| + 2^-24 per element behind:
//       sum_j |x~_j| (...) <= 2^-21 |x||q| + 2^-24 sqrt(d) |x|.
// Everything but (1) is the same for every candidate of a query: bound_common.  (1) is bound_chain(val).  A row NOT in the
// candidate list has an unknown T <= 2 (|q|^2 + max|x|^2): bound_global.  1 % inflation covers the roundings of this
// arithmetic itself and of R^2, |q|^2, |x~| vs |x|.
struct PreBound {
  double common;   // (2) .. (5)
  double chain_k;  // (d + 3) u: the chain's relative error, one u of slack for the 1 / (1 - (d + 2) u)
  double offset;   // val + offset ~ T (squared L2: |q|^2; cosine distance: val + 1 ~ D, no chain term of this form)
  double global;   // bound for a row outside the list
  int metric;
  __host__ __device__ double of(double val) const {  // per-candidate bound
    if (metric) return global;
    double T = val + offset + common;
    T = T > 0.0 ? T : 0.0;
    return common + chain_k * T;
  }
};
// shadow: 0 f32 rows | 1 fp16 shadow, query as fp16 hi + lo | 2 fp16 shadow, query as fp16 hi ONLY (Rq2 = the query's measured
// squared residual |q' - fp16(q')|^2, q' = -2 q or -q as staged: the dropped half contributes |<x~, q' - hi>| <= |x~| sqrt(Rq2),
// |x~| <= |x| + R; and the accumulation makes one pass over the columns instead of two)
__host__ __device__ inline PreBound pre_bound(double qn, double xmax2, double R2, uint32_t d_pad, int metric, int shadow, double Rq2 = 0.0) {
  const double u = 5.9604644775390625e-08, d = (double)d_pad;
  const double qnU = qn * (1.0 + 2.0 * d * u);  // |q|^2 from its rounded sum
  const double S = qnU + xmax2 + (metric ? 1.0 : 0.0);
  const double xm = __builtin_sqrt(xmax2), qm = __builtin_sqrt(qnU);
  PreBound b;
  b.metric = metric;
  double sh = 0.0;
  if (shadow) sh = 1.01 * (2.0 * __builtin_sqrt(R2) * qm + 4.76837158203125e-07 * xm * qm + 5.9604644775390625e-08 * __builtin_sqrt(d) * xm);
  if (shadow == 2) sh += 1.01 * __builtin_sqrt(Rq2 * (1.0 + 2.0 * d * u)) * (xm + __builtin_sqrt(R2));  // (NaN / inf residual: no certificate holds)
  // the round-2 bound, kept for rows outside the list and for the cosine distance: (5 d + 32) u S (+ shadow)
  b.global = (5.0 * d + 32.0) * u * S * (shadow ? 1.01 : 1.0) + sh;
  // (2) (d + 1) u (max|x|^2 + |q|^2)   (3) d u |x| |2 q| (the hi + lo split doubles the accumulation steps, not the partial sums' size:
  // 2 d roundings of u |x||q'| / ... kept at the model's 2 d u |x||q| per pass: x 2 with the shadow)   (4) 16 u S
  const double mf = (shadow == 1 ? 2.0 : 1.0) * 2.0 * d * u * xm * qm;
  b.common = ((d + 1.0) * u * (xmax2 + qnU) + mf + 16.0 * u * S) * 1.01 + sh;
  b.chain_k = (d + 3.0) * u;
  b.offset = qnU;
  if (b.common > b.global) b.common = b.global;  // (never looser than the bound it replaces)
  return b;
}

}  // namespace vers
