// single.hip.h -- the list scans of ONE query (included by ivf_search.hip only): scan1_kernel / scan1t_kernel
// (f32 rows through the ordered chains: a 64-row tile per wave / per block of 16 waves), scan1h_kernel (the inverted lists' fp16 shadow,
// finished by ivf_rescore_kernel<16>: finish.hip.h) and flat1h_kernel (the flat index's shadow).
#pragma once
#include "finish.hip.h"
#include "ivf_src.hip.h"

namespace vers {

template <int METRIC>
__global__ __launch_bounds__(kWave * kWavesPerBlock) void scan1_kernel(Scan1Args a, ScanParams p) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t n_waves = gridDim.x * kWavesPerBlock;
  uint32_t it = blockIdx.x * kWavesPerBlock + wid;
  // (both loads go out together: the record buffer holds at least one entry per launched wave, a stale one is never used)
  const uint32_t n_items = *a.n_items_dev;
  Rec1Src src;
  src.r = a.recs[it];
  src.partials = a.partials; src.k_keep = a.k_keep; src.S_max = a.S_max; src.bound_per_pair = a.bound_per_pair;
  bool nan_seen = false;
  while (it < n_items) {
    ItemView<1> v;
    v.rows = a.rows + (uint64_t)src.r.row0 * p.ld;
    v.nrows = src.r.nrows;
    v.nq = 1;
    v.qb = a.qp;
    scan_item<1, 1, METRIC>(src, p, it, v, lane, nan_seen);
    it += n_waves;
    if (it < n_items) src.r = a.recs[it];
  }
  if (__ballot(nan_seen) != 0 && lane == 0) atomicOr(p.status, 1u);
}

// ---- single query, f32 rows, a tile per BLOCK (round 5) ---------------------------------------------------------------------------
// scan1_kernel gives a 64-row tile to ONE wave, which walks its 192 KiB (d = 768) with 24 KiB in flight: eight dependent round trips.
// That is what bounds the reference's own mode (nprobe = 0: the nearest list -- 38 tiles at cfg3 -- and whatever spills): 23.4 us of a
// 52 us call for 7.5 MB.  Here a tile belongs to a block of 16 waves, exactly as in coarse1_kernel (ivf_plan.hip): wave w loads chunk
// w of every phase of 16 chunks -- the whole tile is in flight at once --, computes its rows' PRODUCTS (x - q)^2 (or x * q), which do
// not depend on the running sum, into LDS; wave 0 walks the strictly ordered chain acc = acc + m_j over them: the same operations
// on the same operands in the same order as scan_item's chain (base.rs:119-126).  Same slots, same merge kernel behind it.
// Blocks stride over the item records (a record = one tile here: plan1_block cuts single queries' lists into 64-row segments).
constexpr int kT1Waves = 16;  // chunks of a phase = waves of the block
constexpr size_t kT1LdsBytes = (size_t)kT1Waves * kLoads * kWave * sizeof(f32x4);  // 128 KiB of products
template <int METRIC>
__global__ __launch_bounds__(kWave * kT1Waves) void scan1t_kernel(Scan1Args a, ScanParams p) {
  extern __shared__ __attribute__((aligned(16))) f32x4 t1_prod[];  // [chunk of the phase][load][lane]
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t n_items = *a.n_items_dev;
  const uint32_t tile_bytes = p.ld * 256u;
  const uint32_t lane_off = (uint32_t)lane * 16u;
  for (uint32_t it = blockIdx.x; it < n_items; it += gridDim.x) {  // (block-uniform)
    const Item1Rec r = a.recs[it];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.rows + (uint64_t)r.row0 * p.ld), 0, (int)tile_bytes, 0x00020000);
    // (chunks past the end of the tile are out of the descriptor's range: they load zeros and are never used)
    auto issue = [&](u32x4 (&b)[kLoads], uint32_t ch) {
#pragma unroll
      for (int i = 0; i < kLoads; ++i) b[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, ch * (kLoads * 1024u) + (uint32_t)i * 1024u, 2);
    };
    auto products = [&](const u32x4 (&b)[kLoads], uint32_t ch) {
      cfloat_as4* qs = (cfloat_as4*)(a.qp + ch * kChunk);
#pragma unroll
      for (int i = 0; i < kLoads; ++i) {
        f32x4 m;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float xv = __uint_as_float(b[i][u]);
          const float sv = qs[i * 4 + u];
          if (METRIC == 0) {
            const float t = __fsub_rn(xv, sv);
            m[u] = __fmul_rn(t, t);
          } else {
            m[u] = __fmul_rn(xv, sv);
          }
        }
        t1_prod[(wid * kLoads + i) * kWave + lane] = m;
      }
    };
    u32x4 bufA[kLoads], bufB[kLoads];
    float acc = 0.0f;
    auto phase = [&](const u32x4 (&cur)[kLoads], u32x4 (&nxt)[kLoads], uint32_t c0) {
      issue(nxt, c0 + kT1Waves + (uint32_t)wid);  // the next phase's chunk: in flight under this phase's chain
      if (c0 + (uint32_t)wid < p.n_chunks) products(cur, c0 + (uint32_t)wid);
      __syncthreads();
      if (wid == 0) {
        const uint32_t nch = p.n_chunks - c0 < (uint32_t)kT1Waves ? p.n_chunks - c0 : (uint32_t)kT1Waves;
        auto ld = [&](f32x4 (&m)[kLoads], uint32_t s) {
#pragma unroll
          for (int i = 0; i < kLoads; ++i) m[i] = t1_prod[(s * kLoads + i) * kWave + lane];
        };
        auto add = [&](const f32x4 (&m)[kLoads]) {
#pragma unroll
          for (int i = 0; i < kLoads; ++i)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __fadd_rn(acc, m[i][u]);
        };
        f32x4 mA[kLoads], mB[kLoads];
        const uint32_t last = nch - 1;
        ld(mA, 0);
        for (uint32_t s = 0; s < nch; s += 2) {
          ld(mB, s + 1 < nch ? s + 1 : last);
          add(mA);
          if (s + 1 < nch) {
            ld(mA, s + 2 < nch ? s + 2 : last);
            add(mB);
          }
        }
      }
      __syncthreads();
    };
    issue(bufA, (uint32_t)wid);
    for (uint32_t c0 = 0; c0 < p.n_chunks; c0 += 2 * kT1Waves) {
      phase(bufA, bufB, c0);
      if (c0 + kT1Waves < p.n_chunks) phase(bufB, bufA, c0 + kT1Waves);
    }
    if (wid == 0) {
      const bool valid = (uint32_t)lane < r.nrows;
      const float dist = METRIC == 0 ? acc : __fsub_rn(1.0f, acc);
      if (__ballot(valid && dist != dist) != 0 && lane == 0) atomicOr(p.status, 1u);
      uint64_t key = valid ? make_key(dist, r.seq0 + (uint32_t)lane) : kKeyMax;
      if (p.lower != nullptr) {  // (wider results than 64 keys come 64 ranks per pass: keys ranked in an earlier pass are dropped)
        const uint64_t lw = p.lower[a.bound_per_pair ? r.out / a.S_max : 0u];
        if (key <= lw) key = kKeyMax;
      }
      wave_rank_sort64(key, lane);  // (unique: they carry their sequence number)
      if (lane < (int)a.k_keep) a.partials[(uint64_t)r.out * a.k_keep + lane] = key;
    }
  }
}

// ---- single query on the fp16 shadow (round 5) ---------------------------------------------------------------------------------
// The ordered-chain scan above streams a query's probed lists as f32 rows: 288 MB at cfg3, 56 us of a 90 us call.  This one streams
// the SHADOW (half the bytes) and does what the batched path does (prescan.hip.h): val = |x|^2 + <x~, q'> (q' = -2 q; cosine: -<x~, q>)
// pre-selects, ivf_rescore_kernel merges the kp smallest vals, certifies, recomputes the survivors in the reference's arithmetic and
// emits; fallback_kernel re-scans exactly when the certificate fails.  Same bits as scan1_kernel + ivf_merge_kernel.
// One query has no use for the matrix cores (a 32x32 tile with one live column): a lane multiplies the 8 fp16 columns its 16-byte
// load delivers -- piece (cb, h) of the shadow tile holds row 32 h + (lane & 31), columns 16 cb + 8 (lane >> 5) .. + 7
// (rows_to_f16_kernel) -- with the f32 query from LDS (v_fma_mix: an fp16 x f32 product is exact in the fma), two independent chains
// per lane (h = 0, 1); lanes l and l ^ 32 hold the two column halves of the same rows and are added at the end: a val per lane.
// The bound is pre_bound's with shadow = 1: the rows' measured residual, and an f32 accumulation of d products in any order.
// A record (plan1_block) is up to 4 tiles of one list: a block, a wave per tile; the waves' 64 keys are sorted and merged pairwise
// through LDS, the kp smallest go to the record's partial slot.
constexpr int kS1hWaves = 4;
struct Scan1hArgs {
  const uint16_t* rows_h; const float* xnorm; const Item1Rec* recs; const uint32_t* n_items_dev; const float* qp;
  uint64_t* partials; uint32_t* qflags; uint32_t ld, kp, metric;
};
inline size_t scan1h_lds_bytes(uint32_t ld) { return (size_t)ld * sizeof(float) + (size_t)kS1hWaves * kWave * sizeof(uint64_t); }
__global__ __launch_bounds__(kWave * kS1hWaves) void scan1h_kernel(Scan1hArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s1h_lds[];
  float* const qs = s1h_lds;                                                               // the scaled query
  uint64_t (*sh)[kWave] = reinterpret_cast<uint64_t(*)[kWave]>(s1h_lds + a.ld);            // [waves][64] sorted keys
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (blockIdx.x >= *a.n_items_dev) return;  // (block-uniform; the record buffer holds an entry per launched block)
  const Item1Rec r = a.recs[blockIdx.x];
  const uint32_t n_tiles = (r.nrows + kWave - 1) / kWave;
  const bool have = (uint32_t)wid < n_tiles;
  // the tile: ld / 8 pieces of 1 KiB, walked in groups of 8 (4 column blocks x 2 row halves), three groups in flight
  constexpr int R = 3, kG = 8;
  const uint32_t n_groups = a.ld / 64u;
  const uint32_t tile_bytes = a.ld * 128u;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.rows_h + (uint64_t)r.row0 * a.ld), 0, (int)(n_tiles * tile_bytes), 0x00020000);
  const uint32_t lane_off = (uint32_t)lane * 16u, tile_off = (uint32_t)wid * tile_bytes;
  u32x4 buf[R][kG];
  auto issue = [&](auto btag, uint32_t g) {
    constexpr int B = decltype(btag)::value;
#pragma unroll
    for (int i = 0; i < kG; ++i) buf[B][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, tile_off + g * (kG * 1024u) + (uint32_t)i * 1024u, 2);
  };
  float xn = 0.0f;
  if (have) {  // (wave-uniform) the first groups fly while the block stages the query
    xn = a.xnorm[(uint64_t)r.row0 + (uint32_t)wid * kWave + lane];
    issue(std::integral_constant<int, 0>{}, 0u);
    if (n_groups > 1) issue(std::integral_constant<int, 1>{}, 1u);
  }
  const float qscale = a.metric ? -1.0f : -2.0f;
  for (uint32_t i = threadIdx.x; i < a.ld / 4u; i += kWave * kS1hWaves) {
    const f32x4 v = reinterpret_cast<const f32x4*>(a.qp)[i];
    reinterpret_cast<f32x4*>(qs)[i] = qscale * v;
  }
  __syncthreads();
  uint64_t key = kKeyMax;
  if (have) {
    float acc0 = 0.0f, acc1 = 0.0f;
    const f32x4* const q4 = reinterpret_cast<const f32x4*>(qs) + 2 * (lane >> 5);
    auto step = [&](auto btag, uint32_t g) {
      constexpr int B = decltype(btag)::value;
      if (g + 2 < n_groups) issue(std::integral_constant<int, (B + 2) % R>{}, g + 2);
      if (g < n_groups) {
#pragma unroll
        for (int c = 0; c < kG / 2; ++c) {
          const f32x4 qa = q4[(g * (kG / 2) + c) * 4], qb = q4[(g * (kG / 2) + c) * 4 + 1];
          const f16x8_t x0 = __builtin_bit_cast(f16x8_t, buf[B][2 * c]), x1 = __builtin_bit_cast(f16x8_t, buf[B][2 * c + 1]);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc0 = __builtin_fmaf((float)x0[u], qa[u], acc0);
            acc1 = __builtin_fmaf((float)x1[u], qa[u], acc1);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc0 = __builtin_fmaf((float)x0[4 + u], qb[u], acc0);
            acc1 = __builtin_fmaf((float)x1[4 + u], qb[u], acc1);
          }
        }
      }
    };
    for (uint32_t g = 0; g < n_groups; g += R) {
      step(std::integral_constant<int, 0>{}, g);
      step(std::integral_constant<int, 1>{}, g + 1);
      step(std::integral_constant<int, 2>{}, g + 2);
    }
    const float t0 = acc0 + __shfl_xor(acc0, 32, kWave), t1 = acc1 + __shfl_xor(acc1, 32, kWave);
    const float dot = lane < 32 ? t0 : t1;  // row `lane` of the tile
    const float val = a.metric ? dot : xn + dot;
    const uint32_t row = (uint32_t)wid * kWave + (uint32_t)lane;
    const bool live = row < r.nrows;
    const bool bad = live && !(__builtin_fabsf(val) < __builtin_inff());
    if (__ballot(bad) != 0 && lane == 0) a.qflags[0] = 1u;  // a non-finite val: the query is re-done exactly (ivf_rescore_kernel)
    if (live && !bad) key = make_key(val, r.seq0 + row);
    wave_rank_sort64(key, lane);
  }
  sh[wid][lane] = key;
  __syncthreads();
#pragma unroll
  for (int s = 1; s < kS1hWaves; s <<= 1) {
    if ((wid & (2 * s - 1)) == 0) {
      wave_merge_sorted64(key, sh[wid + s][lane], lane);
      if (2 * s < kS1hWaves && wid != 0) sh[wid][lane] = key;
    }
    if (2 * s < kS1hWaves) __syncthreads();
  }
  if (wid == 0 && lane < (int)a.kp) a.partials[(uint64_t)r.out * a.kp + lane] = key;
}

// ---- the flat index's single query on ITS shadow (round 5; flat_shadow.hpp) -------------------------------------------------------
// utils::search_exhaustive for one query streams every row: 512 MB at cfg2 (N = 1M, d = 128), 92 us through the ordered chains + 8 us of
// merge.  With a shadow of the flat corpus the same pre-selection / certificate / exact re-score as above applies, the corpus being ONE
// list.  Rows are short there (a 64-row tile of the shadow is 16 KB at d = 128, 15.6 k tiles): the grid is persistent -- two blocks of
// four waves per CU, wave w of W walks tiles w, w + W, ... with its load ring running across tile boundaries -- and every wave keeps
// ONE sorted list of its k + slack smallest keys: a tile's 64 keys are sorted and merged in only when one of them beats the list's
// last key.  The block's four lists are folded into its slot; ivf_rescore_kernel<16> reads the 2 x CUs slots as a flat array.
struct Flat1hArgs {
  const uint16_t* rows_h; const float* xnorm; const float* qp; uint64_t* partials; uint32_t* qflags;
  uint32_t ld, kp, metric, n_rows;
};
__global__ __launch_bounds__(kWave * kS1hWaves) void flat1h_kernel(Flat1hArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s1h_lds[];
  float* const qs = s1h_lds;
  uint64_t (*sh)[kWave] = reinterpret_cast<uint64_t(*)[kWave]>(s1h_lds + a.ld);
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t n_tiles = (a.n_rows + kWave - 1) / kWave;
  const uint32_t W = gridDim.x * kS1hWaves, w0 = blockIdx.x * kS1hWaves + (uint32_t)wid;
  const uint32_t my_tiles = w0 < n_tiles ? (n_tiles - w0 + W - 1) / W : 0u;
  constexpr int R = 3, kG = 8;
  const uint32_t n_groups = a.ld / 64u, tile_bytes = a.ld * 128u;
  const uint32_t n_steps = my_tiles * n_groups;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.rows_h, 0, (int)(n_tiles * tile_bytes), 0x00020000);  // (< 4 GB: flat_shadow_usable)
  const uint32_t lane_off = (uint32_t)lane * 16u;
  u32x4 buf[R][kG];
  uint32_t it = w0, ig = 0;  // the load stream's (tile, group)
  auto issue = [&](auto btag) {
    constexpr int B = decltype(btag)::value;
    const uint32_t off = it * tile_bytes + ig * (kG * 1024u);
#pragma unroll
    for (int i = 0; i < kG; ++i) buf[B][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, off + (uint32_t)i * 1024u, 2);
    if (++ig == n_groups) { ig = 0; it += W; }
  };
  if (n_steps > 0) issue(std::integral_constant<int, 0>{});
  if (n_steps > 1) issue(std::integral_constant<int, 1>{});
  const float qscale = a.metric ? -1.0f : -2.0f;
  for (uint32_t i = threadIdx.x; i < a.ld / 4u; i += kWave * kS1hWaves) {
    const f32x4 v = reinterpret_cast<const f32x4*>(a.qp)[i];
    reinterpret_cast<f32x4*>(qs)[i] = qscale * v;
  }
  __syncthreads();
  uint64_t list = kKeyMax;  // the wave's kp smallest keys so far, ascending over the lanes
  bool bad_any = false;
  {
    float acc0 = 0.0f, acc1 = 0.0f;
    // |x|^2 of the tile in work, requested a tile ahead (loaded where it is used it cost every tile a memory round trip)
    float xn_cur = my_tiles ? a.xnorm[(uint64_t)w0 * kWave + lane] : 0.0f;
    const f32x4* const q4 = reinterpret_cast<const f32x4*>(qs) + 2 * (lane >> 5);
    uint32_t ct = w0, cg = 0;  // the compute stream's (tile, group)
    auto step = [&](auto btag, uint32_t s) {
      constexpr int B = decltype(btag)::value;
      if (s + 2 < n_steps) issue(std::integral_constant<int, (B + 2) % R>{});
      if (s < n_steps) {
#pragma unroll
        for (int c = 0; c < kG / 2; ++c) {
          const f32x4 qa = q4[(cg * (kG / 2) + c) * 4], qb = q4[(cg * (kG / 2) + c) * 4 + 1];
          const f16x8_t x0 = __builtin_bit_cast(f16x8_t, buf[B][2 * c]), x1 = __builtin_bit_cast(f16x8_t, buf[B][2 * c + 1]);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc0 = __builtin_fmaf((float)x0[u], qa[u], acc0);
            acc1 = __builtin_fmaf((float)x1[u], qa[u], acc1);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            acc0 = __builtin_fmaf((float)x0[4 + u], qb[u], acc0);
            acc1 = __builtin_fmaf((float)x1[4 + u], qb[u], acc1);
          }
        }
        if (++cg == n_groups) {  // the tile is complete: a val per lane, folded into the wave's list when any of them can enter it
          const float t0 = acc0 + __shfl_xor(acc0, 32, kWave), t1 = acc1 + __shfl_xor(acc1, 32, kWave);
          const float dot = lane < 32 ? t0 : t1;
          const uint32_t row = ct * kWave + (uint32_t)lane;
          const float val = a.metric ? dot : xn_cur + dot;  // (the last tile's padding rows exist in xnorm)
          {
            const uint32_t nt = ct + W < n_tiles ? ct + W : ct;
            xn_cur = a.xnorm[(uint64_t)nt * kWave + lane];
          }
          const bool live = row < a.n_rows;
          const bool bad = live && !(__builtin_fabsf(val) < __builtin_inff());
          bad_any |= bad;
          uint64_t key = live && !bad ? make_key(val, row) : kKeyMax;
          const uint64_t last = readlane64(list, (int)a.kp - 1);
          if (__ballot(key < last) != 0) {  // (wave-uniform)
            wave_rank_sort64(key, lane);
            wave_merge_sorted64(list, key, lane);
          }
          acc0 = acc1 = 0.0f; cg = 0; ct += W;
        }
      }
    };
    for (uint32_t s = 0; s < n_steps; s += R) {
      step(std::integral_constant<int, 0>{}, s);
      step(std::integral_constant<int, 1>{}, s + 1);
      step(std::integral_constant<int, 2>{}, s + 2);
    }
  }
  if (__ballot(bad_any) != 0 && lane == 0) a.qflags[0] = 1u;
  sh[wid][lane] = list;
  __syncthreads();
#pragma unroll
  for (int s = 1; s < kS1hWaves; s <<= 1) {
    if ((wid & (2 * s - 1)) == 0) {
      wave_merge_sorted64(list, sh[wid + s][lane], lane);
      if (2 * s < kS1hWaves && wid != 0) sh[wid][lane] = list;
    }
    if (2 * s < kS1hWaves) __syncthreads();
  }
  if (wid == 0 && lane < (int)a.kp) a.partials[(uint64_t)blockIdx.x * a.kp + lane] = list;
}

}  // namespace vers
