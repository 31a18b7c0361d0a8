// scan.cuh -- the exact-order distance engine and wave top-k shared by every
// kernel of the path (flat scan, coarse quantiser, inverted-list scan, k-means
// assign).  gfx950 only.
//
// Arithmetic contract (the reason this file exists): a distance is
//     acc = 0; for j in 0..d: t = x[j] - q[j]; acc = acc + t*t
// evaluated strictly left to right in f32 with separately rounded multiply and
// add (vers base.rs:119-126; Rust never contracts or re-associates).  Each
// LANE owns one corpus row and walks its columns in order, so the result is
// bit-identical to the reference; the 64 rows of a wave are transposed through
// a private LDS tile so that HBM is still read with full-line coalesced loads.
#pragma once
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
// value of lane-1 (lane 0 receives `fill`)
__device__ __forceinline__ uint64_t shift_up1_64(uint64_t v, uint64_t fill, int lane) {
  uint32_t lo = __shfl_up((uint32_t)v, 1, 64);
  uint32_t hi = __shfl_up((uint32_t)(v >> 32), 1, 64);
  uint64_t r = ((uint64_t)hi << 32) | lo;
  return lane == 0 ? fill : r;
}

// Sorted list of the 64 smallest keys seen so far, one per lane (ascending by
// lane).  `cand` is one candidate per lane (kKeyMax = none).  Only candidates
// below the current k-th key are inserted; keys are unique (seq differs), so
// the order is total and the result equals a stable sort by (dist, seq).
__device__ __forceinline__ void wave_topk_update(uint64_t& list, uint32_t k, uint64_t cand, int lane) {
  uint64_t thr = readlane64(list, (int)k - 1);
  uint64_t m = __ballot(cand < thr);
  while (m) {
    int src = __ffsll((unsigned long long)m) - 1;
    m &= m - 1;
    uint64_t x = readlane64(cand, src);
    if (x < thr) {
      uint64_t prev = shift_up1_64(list, 0, lane);
      uint64_t mx = prev > x ? prev : x;
      list = x < list ? mx : list;
      thr = readlane64(list, (int)k - 1);
    }
  }
}

// ---- the tile engine ---------------------------------------------------------
// One wave, one tile of up to 64 rows (lane == row), QG queries at once.
//   rows      first row of the item (uniform), pitch ld floats (multiple of 4)
//   row0      first row of this tile inside the item
//   nrows     rows in the item (rows >= nrows contribute zeros and are ignored)
//   q[QG]     uniform query pointers, zero padded to n_chunks*64 floats
//   tile      this wave's private LDS tile [64][kLdsStride]
// METRIC 0: acc = sum (x-q)^2      METRIC 1: acc = sum x*q   (caller does 1-acc)
struct TileLoader {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t ld_bytes;      // row pitch in bytes
  uint32_t voff_lane;     // (lane>>4)*ld_bytes + (lane&15)*16
  uint32_t col_byte;      // (lane&15)*16
  uint32_t lds_write_off; // float index inside the tile for i = 0

  __device__ __forceinline__ void init(const float* rows, uint64_t item_bytes, uint32_t ld, int lane) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)rows, 0, (int)(uint32_t)item_bytes, 0x00020000);
    ld_bytes = ld * 4u;
    col_byte = (uint32_t)(lane & 15) * 16u;
    voff_lane = (uint32_t)(lane >> 4) * ld_bytes + col_byte;
    lds_write_off = (uint32_t)(lane >> 4) * kLdsStride + (uint32_t)(lane & 15) * 4u;
  }
  // issue the 16 loads of (tile row0, chunk c); rows past the item read 0 (buffer bounds check),
  // columns past the pitch are zeroed in stage() -- NOT here, a select on the loaded value would
  // make the compiler wait for the load right after issuing it and kill the prefetch.
  __device__ __forceinline__ void issue(u32x4 (&r)[16], uint32_t row0, uint32_t c) const {
    const uint32_t chunk_byte = c * (kChunk * 4u);
    const uint32_t voff = voff_lane + chunk_byte;
    const uint32_t sbase = row0 * ld_bytes;  // uniform
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      // the row offset must sit in voffset: soffset is not part of the hardware range check
      r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + (sbase + (uint32_t)i * 4u * ld_bytes), 0, 0);
    }
  }
  __device__ __forceinline__ void stage(const u32x4 (&r)[16], float* tile, uint32_t c) const {
    const bool col_ok = c * (kChunk * 4u) + col_byte < ld_bytes;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      *reinterpret_cast<u32x4*>(tile + lds_write_off + i * 4 * kLdsStride) = col_ok ? r[i] : u32x4{0u, 0u, 0u, 0u};
  }
};

// Query operands are wave-uniform and come in through the scalar path.  For QG > 1 the
// item's queries are stored INTERLEAVED, qb[col * QG + qi], so that one s_load_dwordx8
// brings element `col` of all 8 queries and adjacent SGPR pairs feed v_pk_*_f32:
// two queries per VALU instruction, each still its own strictly ordered f32 chain.
template <int QG, int METRIC>
__device__ __forceinline__ void tile_chunk_compute(f32x2 (&acc)[(QG + 1) / 2], const float* tile, int lane,
                                                   const float* qb, uint32_t c) {
  const float* myrow = tile + lane * kLdsStride;
  if constexpr (QG == 1) {
    cfloat_as4* qs = (cfloat_as4*)(qb + c * kChunk);
    float a = acc[0][0];
#pragma unroll
    for (int j16 = 0; j16 < 4; ++j16) {
      f32x4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const f32x4*>(myrow + j16 * 16 + u * 4);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        float xv = x[u >> 2][u & 3];
        float sv = qs[j16 * 16 + u];
        if (METRIC == 0) {
          float t = __fsub_rn(xv, sv);
          a = __fadd_rn(a, __fmul_rn(t, t));
        } else {
          a = __fadd_rn(a, __fmul_rn(xv, sv));
        }
      }
    }
    acc[0][0] = a;
  } else {
    typedef __attribute__((address_space(4))) const f32x2 cf32x2_as4;
    cf32x2_as4* qs = (cf32x2_as4*)(qb + (size_t)c * kChunk * QG);
    // rolled over groups of 4 columns: unrolling the whole chunk lets the scheduler hoist every
    // scalar load (64 * QG SGPRs) and spill the SGPR file.
#pragma unroll 1
    for (int j4 = 0; j4 < kChunk / 4; ++j4) {
      f32x4 x = *reinterpret_cast<const f32x4*>(myrow + j4 * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const f32x2 xx = {x[u], x[u]};
#pragma unroll
        for (int p = 0; p < QG / 2; ++p) {
          const f32x2 sv = qs[(j4 * 4 + u) * (QG / 2) + p];
          if (METRIC == 0) {
            const f32x2 t = xx - sv;
            acc[p] = acc[p] + t * t;
          } else {
            acc[p] = acc[p] + xx * sv;
          }
        }
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
//   uint32_t n_items() const
//   void     get(it, ItemView<QG>&) const
//   uint32_t seq_base(it, qi) const          seq of the item's first row for query qi
//   uint64_t* out(it, qi) const              partial slot (k keys) for query qi
//   static constexpr bool kSeqIds            seq = seq_ids(it)[row] instead of seq_base + row
//   const uint32_t* seq_ids(it) const
template <int QG>
struct ItemView {
  const float* rows;   // first row of the item
  uint32_t nrows;      // rows in the item
  uint32_t nq;         // live queries (<= QG)
  const float* qb;     // QG == 1: the query; else the item's interleaved query block [col][QG]
};                     // (zero padded to n_chunks*64 columns; dead query slots are zeros)

struct ScanParams {
  uint32_t ld;        // row pitch in floats (multiple of 4)
  uint32_t n_chunks;  // ceil(ld / 64)
  uint32_t k;         // keys kept per query (<= 64)
  uint32_t* status;   // device word: bit0 = NaN seen
};

template <int QG, int METRIC, class Src>
__global__ __launch_bounds__(kWave * kWavesPerBlock) void scan_kernel(Src src, ScanParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* tile = lds + wid * (kWave * kLdsStride);
  const uint32_t n_waves = gridDim.x * kWavesPerBlock;
  const uint32_t n_items = src.n_items();
  bool nan_seen = false;

  for (uint32_t it = blockIdx.x * kWavesPerBlock + wid; it < n_items; it += n_waves) {
    ItemView<QG> v;
    src.get(it, v);
    uint64_t list[QG];
#pragma unroll
    for (int qi = 0; qi < QG; ++qi) list[qi] = kKeyMax;

    TileLoader L;
    L.init(v.rows, (uint64_t)v.nrows * p.ld * 4u, p.ld, lane);
    const uint32_t n_tiles = (v.nrows + kWave - 1) / kWave;
    const uint32_t n_steps = n_tiles * p.n_chunks;
    u32x4 r[16];
    uint32_t sid = 0;
    if (Src::kSeqIds && lane < (int)v.nrows) sid = src.seq_ids(it)[lane];  // older than every prefetch below
    L.issue(r, 0, 0);
    f32x2 acc[(QG + 1) / 2];
#pragma unroll
    for (int p2 = 0; p2 < (QG + 1) / 2; ++p2) acc[p2] = f32x2{0.0f, 0.0f};
    uint32_t t = 0, c = 0;
    for (uint32_t s = 0; s < n_steps; ++s) {
      L.stage(r, tile, c);
      uint32_t tn = t, cn = c + 1;
      if (cn == p.n_chunks) { cn = 0; tn = t + 1; }
      uint32_t sid_next = 0;
      if (Src::kSeqIds && cn == 0 && tn * kWave + lane < v.nrows) sid_next = src.seq_ids(it)[tn * kWave + lane];
      if (s + 1 < n_steps) L.issue(r, tn * kWave, cn);  // prefetch the next step under this step's math
      tile_chunk_compute<QG, METRIC>(acc, tile, lane, v.qb, c);
      if (cn == 0) {  // tile finished: fold its 64 candidates into the per-query lists
        const uint32_t row = t * kWave + lane;
        // kSeqIds: rows whose id is 0xFFFFFFFF are unused storage slack, not vectors
        const bool valid = row < v.nrows && (!Src::kSeqIds || sid != 0xFFFFFFFFu);
#pragma unroll
        for (int qi = 0; qi < QG; ++qi) {
          if (qi < (int)v.nq) {
            float dist = METRIC == 0 ? acc[qi >> 1][qi & 1] : __fsub_rn(1.0f, acc[qi >> 1][qi & 1]);
            nan_seen |= valid && (dist != dist);
            uint32_t seq = Src::kSeqIds ? sid : src.seq_base(it, qi) + row;
            uint64_t cand = valid ? make_key(dist, seq) : kKeyMax;
            wave_topk_update(list[qi], p.k, cand, lane);
          }
          acc[qi >> 1][qi & 1] = 0.0f;
        }
        sid = sid_next;
      }
      t = tn; c = cn;
    }
#pragma unroll
    for (int qi = 0; qi < QG; ++qi)
      if (qi < (int)v.nq && lane < (int)p.k) src.out(it, qi)[lane] = list[qi];
  }
  if (__ballot(nan_seen) != 0 && lane == 0) atomicOr(p.status, 1u);
}

// ---- merge of partial slots ----------------------------------------------------
// One BLOCK of kMergeWaves waves per output group folds `n_keys` keys (partial slots,
// kKeyMax padded) into the top-k: every wave folds a strided share with several
// independent loads in flight (the loop is latency-bound otherwise), wave 0 then folds
// the per-wave lists through LDS.  Returns the final list in wave 0 (other waves: junk).
constexpr int kMergeWaves = 16;
__device__ __forceinline__ uint64_t block_merge_keys(const uint64_t* keys, uint32_t n_keys, uint32_t k,
                                                     uint64_t (*sh)[kWave]) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int U = 4;
  uint64_t list = kKeyMax;
  for (uint32_t base = wid * kWave; base < n_keys; base += kMergeWaves * kWave * U) {
    uint64_t cand[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t i = base + u * (kMergeWaves * kWave) + lane;
      cand[u] = i < n_keys ? keys[i] : kKeyMax;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) wave_topk_update(list, k, cand[u], lane);
  }
  __syncthreads();  // `sh` may still be read by wave 0 of a previous call
  sh[wid][lane] = list;
  __syncthreads();
  if (wid == 0) {
    for (int w = 1; w < kMergeWaves; ++w) wave_topk_update(list, k, sh[w][lane], lane);
  }
  return list;
}

}  // namespace vers
