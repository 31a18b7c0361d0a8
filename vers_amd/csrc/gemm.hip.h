// gemm.hip.h -- the batched query x centroid contraction of the coarse quantiser on the f32 matrix
// cores (v_mfma_f32_32x32x2_f32), with exact re-scoring and a certificate.
//
// The reference ranks centroids by the ordered f32 sum of (c-q)^2 (ivfflat.rs:155-161).  A GEMM
// cannot reproduce that rounding, so it is used ONLY to pre-select: for a batch of queries
//     G[m][n] = |c_n|^2 - 2 <q_m, c_n>                 (f32 MFMA, 1024 x 4096 x 768 at cfg3)
// approximates D(c_n, q_m) - |q_m|^2.  Per query the P+S smallest G are taken (S = slack), their
// distances are re-computed in the reference's own arithmetic (one lane per candidate, ordered
// chain), sorted by the exact (distance, index) key, and the result is CERTIFIED: with
// tau = largest selected G and E a rigorous bound on |G + |q|^2 - D_ref| (both roundings, below),
// every unselected centroid has D_ref >= tau + |q|^2 - E; if the P-th exact distance is below that,
// no unselected centroid can be among the true top-P and the output equals the exact coarse
// quantiser bit for bit.  A query that fails the certificate is re-done exactly by its own wave
// (all centroids, ordered chains) -- rare, and never wrong.
//
// Error bound (u = 2^-24, d = padded length, S = |q|^2 + max|c|^2, all sums of non-negative terms
// or Cauchy-Schwarz):  |D_ref - T| <= (d+2) u T <= 2(d+2) u S ;  |c|^2, |q|^2 by ordered sums:
// <= d u |.|^2 each ;  MFMA dot = k-ordered fma chain: <= d u |q||c| <= d u S / 2, doubled ;
// two final roundings <= 2 u (3S).   Total <= (5d + 16) u S =: E.
#pragma once
#include <cstdlib>

#include "plan.hip.h"
#include "scan.hip.h"
#include "staged.hip.h"
#include "wide.hip.h"

namespace vers {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

constexpr int kGemmBM = 128, kGemmBN = 128, kGemmBK = 32;
constexpr int kGemmLds = kGemmBK + 4;  // row pitch in floats: 4*odd -> conflict-free ds_read_b128 by row

// |c_n|^2 (ordered sum; any rounding is covered by E) for n < k, +inf for padding rows
static __global__ void row_norms_kernel(const float* C, uint32_t ld, uint32_t k, uint32_t k_pad, float* out) {
  const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= k_pad) return;
  if (n >= k) { out[n] = __builtin_inff(); return; }
  const f32x4* p = reinterpret_cast<const f32x4*>(C + (uint64_t)n * ld);
  float acc = 0.0f;
  for (uint32_t j = 0; j < ld / 4; ++j) {
    const f32x4 v = p[j];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __fadd_rn(acc, __fmul_rn(v[u], v[u]));
  }
  out[n] = acc;
}

// Block -> tile.  The grid is one-dimensional and the hardware deals consecutive workgroups round-robin over the 8 XCDs,
// each with its own 4 MB L2.  grp == 0: column tile fastest (a small problem that is resident as a whole: the coarse
// quantiser).  grp > 0 (needs m_tiles % (8 * grp) == 0): XCD x owns the row tiles [x * m_tiles/8, (x+1) * m_tiles/8) and
// walks them in chunks of grp tiles (grp * 128 rows of the M operand = 1.5 MB at K = 768 stay in ITS L2) against every
// column tile in turn, so a column tile is fetched once per chunk and XCD -- and because the XCDs advance in step, one
// of those 8 fetches comes from HBM and seven from the Infinity Cache.  The k-means assign pass (M = centroids, N = the
// batch of points, 400 MB) with the plain order re-read the points from HBM once per centroid tile: 32 x at k = 4096.
__device__ __forceinline__ void gemm_tile_coords(uint32_t m_tiles, uint32_t n_tiles, uint32_t grp, uint32_t& tm, uint32_t& tn, uint32_t L = blockIdx.x) {
  if (grp == 0) { tn = L % n_tiles; tm = L / n_tiles; return; }
  const uint32_t xcd = L & 7u, j = L >> 3, per_xcd = m_tiles >> 3;
  const uint32_t chunk = j / (grp * n_tiles), w = j % (grp * n_tiles);
  tn = w / grp;
  tm = xcd * per_xcd + chunk * grp + w % grp;
}
inline uint32_t gemm_tile_group(uint32_t m_tiles) {  // host side: the chunk for an M operand that is worth keeping in L2
  for (uint32_t g : {4u, 2u, 1u}) if (m_tiles % (8u * g) == 0) return g;
  return 0;
}

// Epilogue shared by the two contraction kernels (the C/D layout of the 32x32 MFMAs does not depend on the input type):
// col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).  `lds` is the block's (dead) operand storage.
template <bool NORM_ROWS>
__device__ __forceinline__ void gemm_epilogue(f32x16 (&acc)[2][2], float* lds, const float* __restrict__ cnorm, uint32_t N_pad,
                                              float* __restrict__ G, int metric, uint32_t k_rows, float* __restrict__ part_v1,
                                              uint32_t* __restrict__ part_c1, float* __restrict__ part_v2, uint32_t m0, uint32_t n0, int wr,
                                              int wc, int r, int hh) {
  float* As = lds;
  // epilogue: C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  if constexpr (!NORM_ROWS) {
    // (64 `global_store_dword` of two 128-byte pieces each.  The same values through LDS and out as 16 `global_store_dwordx4` of
    // whole 256-byte row segments were tried in round 4: 27.4 vs 27.6 us -- the 7 us this epilogue costs are the 16 MB of G on
    // their way to memory, not the store instructions.)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const uint32_t n = n0 + wc * 64 + b * 32 + r;
        const float cn = metric ? 0.0f : cnorm[n];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const uint32_t m = m0 + wr * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          G[(uint64_t)m * N_pad + n] = metric ? -acc[a][b][e] : cn - 2.0f * acc[a][b][e];
        }
      }
  } else {
    // per point: (smallest value, its centroid, second smallest) over this tile's centroids.  A tie with the candidate makes
    // second == best, a NaN makes second NaN: neither certifies (assign_rescore_kernel) and the exact scan decides --
    // so the order in which equal values meet does not matter here.
    auto fold = [](float& v1, uint32_t& c1, float& v2, float w1, uint32_t d1, float w2) {
      const bool nan = (v2 != v2) || (w2 != w2);
      if (w1 < v1) { v2 = v1 < w2 ? v1 : w2; v1 = w1; c1 = d1; }
      else { const float t = w1 < v2 ? w1 : v2; v2 = t; }  // includes w1 == v1: second == best
      if (nan) v2 = __builtin_nanf("");
    };
    // (the tile's 128 centroid norms through LDS, four consecutive rows per ds_read_b128: wide_epilogue's note)
    __syncthreads();  // the operand tiles are dead
    float* const cn_s = As + 512;  // (behind the exchange area below: 384 floats)
    if (threadIdx.x < (unsigned)kGemmBM) cn_s[threadIdx.x] = cnorm[m0 + threadIdx.x];  // (k_pad entries, +inf in the padding)
    __syncthreads();
    f32x4 cn4[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) cn4[a][q] = *reinterpret_cast<const f32x4*>(cn_s + wr * 64 + a * 32 + 8 * q + 4 * hh);
    float bv1[2], bv2[2];
    uint32_t bc1[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float v1 = __builtin_inff(), v2 = __builtin_inff();
      uint32_t c1 = m0 + wr * 64 + 4 * hh;
      bool nan = false;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const uint32_t m = m0 + wr * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          float g = metric ? -acc[a][b][e] : cn4[a][e >> 2][e & 3] - 2.0f * acc[a][b][e];
          if (m >= k_rows) g = __builtin_inff();  // padding centroids (zero rows) are not candidates
          nan |= g != g;
          if (g < v1) { v2 = v1; v1 = g; c1 = m; }
          else if (g < v2 || g == v1) v2 = g;
        }
      if (nan) v2 = __builtin_nanf("");
      // the other half of the rows sits in lane ^ 32
      const float w1 = __shfl_xor(v1, 32, kWave), w2 = __shfl_xor(v2, 32, kWave);
      const uint32_t d1 = (uint32_t)__shfl_xor((int)c1, 32, kWave);
      fold(v1, c1, v2, w1, d1, w2);
      bv1[b] = v1; bc1[b] = c1; bv2[b] = v2;
    }
    // rows 64..127 of the tile belong to the waves wr == 1: through LDS (the operand tiles are dead by now)
    __syncthreads();
    float* xs = As;  // [wc][b][r][3]
    if (wr == 1 && hh == 0) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float* t = xs + ((wc * 2 + b) * 32 + r) * 3;
        t[0] = bv1[b]; t[1] = __uint_as_float(bc1[b]); t[2] = bv2[b];
      }
    }
    __syncthreads();
    if (wr == 0 && hh == 0) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float* t = xs + ((wc * 2 + b) * 32 + r) * 3;
        fold(bv1[b], bc1[b], bv2[b], t[0], __float_as_uint(t[1]), t[2]);
        const uint64_t o = (uint64_t)(m0 / kGemmBM) * N_pad + n0 + wc * 64 + b * 32 + r;
        part_v1[o] = bv1[b]; part_c1[o] = bc1[b]; part_v2[o] = bv2[b];
      }
    }
  }
}


// G[m][n] = cnorm[n] - 2 * sum_k Q[m][k] * C[n][k].   Q [M_pad][K], C [N_pad][K] row-major, K % 32 == 0,
// M_pad % 128 == 0, N_pad % 128 == 0.  Block = 4 waves, 128 x 128 tile; wave = 64 x 64 (2 x 2 MFMA tiles).
// LDS tiles hold k permuted as [row][h = k & 1][s = k >> 1] so that lane (r, h) reads its 16 operands
// of a K-tile (k = 2s + h) as four contiguous ds_read_b128.
// NORM_ROWS = false: G[m][n] = norm[n] - 2 dot (coarse quantiser: m = query, n = centroid);
// NORM_ROWS = true : G[m][n] = norm[m] - 2 dot (k-means assign: m = centroid, n = point, so that a point's
//                    values are a coalesced column walk for the per-point selection).
// metric 1 (cosine distance 1 - dot, base.rs:153-155): G = -dot, which approximates D_ref - 1; the padding rows /
// columns of the operands are zero, so padded entries come out as 0 (the selections only look at real ones).
// NORM_ROWS = true does not write G at all: its epilogue reduces the block's 128 x 128 tile to, per point (column),
// the smallest value, its centroid and the second smallest over the tile's 128 centroids (rows m < k_rows) and writes
// that triple to part_*[row tile][n] -- 12 bytes per (row tile, point) instead of 512; assign_argmin_merge_kernel
// folds the k/128 triples of a point.  (Round 1 wrote Gt -- 2.1 GB per 131072-point batch at k = 4096 -- and read it
// back in a separate arg-min kernel.)
template <bool NORM_ROWS>
static __global__ __launch_bounds__(256) void dist_gemm_kernel(const float* __restrict__ Q, const float* __restrict__ C,
                                                               const float* __restrict__ cnorm, uint32_t K, uint32_t N_pad,
                                                               float* __restrict__ G, int metric, uint32_t m_tiles, uint32_t n_tiles, uint32_t grp, uint32_t k_rows = 0,
                                                               float* __restrict__ part_v1 = nullptr, uint32_t* __restrict__ part_c1 = nullptr,
                                                               float* __restrict__ part_v2 = nullptr) {
  __shared__ __attribute__((aligned(16))) float As[kGemmBM * kGemmLds];
  __shared__ __attribute__((aligned(16))) float Bs[kGemmBN * kGemmLds];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;  // wave position inside the block tile
  uint32_t tile_m, tile_n;
  gemm_tile_coords(m_tiles, n_tiles, grp, tile_m, tile_n);
  const uint32_t m0 = tile_m * kGemmBM, n0 = tile_n * kGemmBN;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

  // staging map: 128 rows x 8 float4 per matrix = 1024 float4, 4 per thread
  const int srow = tid >> 3, sc4 = tid & 7;  // rows srow + 32*i
  f32x4 ra[4], rb[4];
  auto gload = [&](uint32_t k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const f32x4*>(Q + (uint64_t)(m0 + srow + 32 * i) * K + k0 + sc4 * 4);
      rb[i] = *reinterpret_cast<const f32x4*>(C + (uint64_t)(n0 + srow + 32 * i) * K + k0 + sc4 * 4);
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // k = 4*sc4 + {0,1,2,3} -> (h, s) = (0, 2*sc4), (1, 2*sc4), (0, 2*sc4+1), (1, 2*sc4+1)
      float* a = As + (srow + 32 * i) * kGemmLds + 2 * sc4;
      float* b = Bs + (srow + 32 * i) * kGemmLds + 2 * sc4;
      *reinterpret_cast<f32x2*>(a) = f32x2{ra[i][0], ra[i][2]};
      *reinterpret_cast<f32x2*>(a + 16) = f32x2{ra[i][1], ra[i][3]};
      *reinterpret_cast<f32x2*>(b) = f32x2{rb[i][0], rb[i][2]};
      *reinterpret_cast<f32x2*>(b + 16) = f32x2{rb[i][1], rb[i][3]};
    }
  };
  const int r = lane & 31, hh = lane >> 5;
  gload(0);
  for (uint32_t k0 = 0; k0 < K; k0 += kGemmBK) {
    __syncthreads();  // previous tile's readers done
    lstore();
    __syncthreads();
    if (k0 + kGemmBK < K) gload(k0 + kGemmBK);  // next tile in flight under the MFMAs
    f32x4 fa[2][4], fb[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int v4 = 0; v4 < 4; ++v4) {
        fa[t][v4] = *reinterpret_cast<const f32x4*>(As + (wr * 64 + t * 32 + r) * kGemmLds + hh * 16 + v4 * 4);
        fb[t][v4] = *reinterpret_cast<const f32x4*>(Bs + (wc * 64 + t * 32 + r) * kGemmLds + hh * 16 + v4 * 4);
      }
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][s >> 2][s & 3], fb[b][s >> 2][s & 3], acc[a][b], 0, 0, 0);
  }
  gemm_epilogue<NORM_ROWS>(acc, As, cnorm, N_pad, G, metric, k_rows, part_v1, part_c1, part_v2, m0, n0, wr, wc, r, hh);
}

// ---- the same contraction as THREE bf16 MFMA products of hi/lo-split operands -----------------------------------------
// x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi) (both round-to-nearest; x - hi is exact), |r| <= 2^-18 |x|.
//     <a, b> ~ <a_hi, b_hi> + <a_hi, b_lo> + <a_lo, b_hi>          (v_mfma_f32_32x32x16_bf16, f32 accumulation)
// drops <a_lo, b_lo> and the residuals: |error| <= 3 * 2^-18 |a||b| on top of the f32 accumulation error that the f32
// kernel has too -- 1.1e-5 |a||b| against a certificate bound E of 2.3e-4 (|a|^2 + |b|^2) at d = 768.  The consumers add
// kX3Slack * (|q|^2 + max|c|^2) to their E (coarse_select_rescore_kernel, assign_rescore_kernel), so the product is a
// PRE-FILTER exactly like the f32 one: results stay the reference's bits.  The matrix cores run bf16 at 16x the f32
// rate, three products = 5.3x fewer MFMA cycles; the kernel is then bound by its operand traffic (global -> split in
// registers -> LDS -> fragments), not by the MFMAs.
constexpr float kX3Slack = 1.6e-5f;  // >= 2 * 3 * 2^-18 (G = norm - 2 dot doubles the dot's error), rounded up
// ONE product (dist_gemm_x3w_kernel<2, 1>, the first filter of the assign cascade) on FP16 operands x~ = fp16(x), c~ = fp16(c): the products
// are exact in the f32 accumulator, so <x, c> - <x~, c~> = <x - x~, c> + <x~, c - c~>, bounded with the MEASURED residuals r_x = |x - x~|
// (per point, summed by assign_rescore_kernel next to |x|^2) and R_c = max |c - c~| (to_f16_resid_kernel): |.| <= r_x |c| + (|x| + r_x) R_c;
// G doubles it.  A bf16 single product was tried first: 2^-8 (|x|^2 + |c|^2) = 7e-3 at unit norms left most points of cfg3's corpus open
// (the pass 3x slower); fp16's 11 bits bring the window to ~1e-3, twice the three-product filter's.  Elements beyond fp16's range make
// the residual infinite: nothing certifies, the exact paths decide.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 gf16x8;  // (the single-product first filter of the assign cascade runs on fp16)
typedef __attribute__((ext_vector_type(4))) _Float16 gf16x4;
// LDS rows are the 32 bf16 of a K-tile, 64 bytes, unpadded; the four 16-byte chunks of row r sit at chunk ^ ((r >> 2) & 3).
// A ds_read_b128 is served in groups of 16 lanes = 16 rows with r % 4 and (r >> 2) % 4 covering all 16 combinations, so the
// group touches all 64 banks once; the 8-byte split stores of 16 consecutive threads cover two whole rows = 32 banks once.
// (The first cut padded rows to 80 bytes: conflict-free reads, but every split store 2-way conflicted -- a third of the
// kernel's LDS cycles by SQ_LDS_BANK_CONFLICT.)
constexpr int kX3Pitch = 32;
__device__ __forceinline__ int x3_chunk(int row, int chunk) { return (chunk ^ ((row >> 2) & 3)) * 8; }  // bf16 offset inside the row
constexpr size_t kX3LdsBytes = 2 * 2 * 2 * (size_t)kGemmBM * kX3Pitch * 2;  // two buffers x A|B x hi|lo = 64 KB: two blocks per CU

// x -> bf16 hi, lo (both round-to-nearest; the split the kernel below otherwise does on the fly), for an operand that many
// blocks re-read: the centroids -- M operand of the k-means assign pass (split once per pass), N operand of the coarse
// quantiser (split when the index is installed).  PRE bit 0: the M operand comes pre-split (Qh, Ql), bit 1: the N operand.
static __global__ void split_bf16_kernel(const float* __restrict__ x, uint64_t n4, __bf16* __restrict__ hi, __bf16* __restrict__ lo) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
  bf16x4 h, l;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    h[u] = (__bf16)v[u];
    l[u] = (__bf16)(v[u] - (float)h[u]);
  }
  reinterpret_cast<bf16x4*>(hi)[i] = h;
  reinterpret_cast<bf16x4*>(lo)[i] = l;
}

template <bool NORM_ROWS, int PRE>
static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void dist_gemm_x3_kernel(const float* __restrict__ Q, const float* __restrict__ C,
                                                                  const __bf16* __restrict__ Qh, const __bf16* __restrict__ Ql,
                                                                  const __bf16* __restrict__ Ch, const __bf16* __restrict__ Cl,
                                                                  const float* __restrict__ cnorm, uint32_t K, uint32_t N_pad,
                                                                  float* __restrict__ G, int metric, uint32_t m_tiles, uint32_t n_tiles, uint32_t grp, uint32_t k_rows = 0,
                                                                  float* __restrict__ part_v1 = nullptr, uint32_t* __restrict__ part_c1 = nullptr,
                                                                  float* __restrict__ part_v2 = nullptr, uint32_t* __restrict__ zero = nullptr,
                                                                  uint32_t n_zero = 0) {
  // (the batch's zero-initialised planning tables, for the launch behind this one: a few words per thread here instead of a
  // memset launch of their own in front of every batch's coarse quantiser)
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_zero; i += gridDim.x * 256u) zero[i] = 0u;
  // [buffer][matrix A|B][part hi|lo][128 rows][kX3Pitch] bf16 = 2 x 32 KB: the split tile of step k+1 is written while the
  // MFMAs of step k read the other buffer (ONE barrier per K-tile), and the global loads run TWO tiles ahead in
  // registers -- with the MFMA time of a tile down to ~770 cycles a single tile of prefetch no longer covers the L2 /
  // Infinity-Cache latency (the first cut of this kernel, single-buffered with one tile of prefetch: 3.83 ms per
  // 131072 x 4096 x 768 assign batch).
  extern __shared__ __attribute__((aligned(16))) __bf16 T[];
  constexpr int kPart = kGemmBM * kX3Pitch;  // one [128][kX3Pitch] array
  auto Tp = [&](int buf, int mat, int part) { return T + ((buf * 2 + mat) * 2 + part) * kPart; };
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  uint32_t tile_m, tile_n;
  gemm_tile_coords(m_tiles, n_tiles, grp, tile_m, tile_n);
  const uint32_t m0 = tile_m * kGemmBM, n0 = tile_n * kGemmBN;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;
  const int srow = tid >> 3, sc4 = tid & 7;  // rows srow + 32*i, float4 column sc4
  // an f32 operand: 4 float4 per thread and tile; a pre-split one: 2 x 16 bytes of hi ([0], [1]) and of lo ([2], [3]) --
  // thread t serves (row, 16-byte chunk) = ((t + 256 i) >> 2, (t + 256 i) & 3), i = 0, 1
  f32x4 ra[2][4], rb[2][4];
  auto gload = [&](auto stag, uint32_t k0) {
    constexpr int S = decltype(stag)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (!(PRE & 1)) ra[S][i] = *reinterpret_cast<const f32x4*>(Q + (uint64_t)(m0 + srow + 32 * i) * K + k0 + sc4 * 4);
      if constexpr (!(PRE & 2)) rb[S][i] = *reinterpret_cast<const f32x4*>(C + (uint64_t)(n0 + srow + 32 * i) * K + k0 + sc4 * 4);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      if constexpr (PRE & 1) {
        const uint64_t at = (uint64_t)(m0 + (idx >> 2)) * K + k0 + (idx & 3) * 8;
        ra[S][i] = *reinterpret_cast<const f32x4*>(Qh + at);
        ra[S][2 + i] = *reinterpret_cast<const f32x4*>(Ql + at);
      }
      if constexpr (PRE & 2) {
        const uint64_t at = (uint64_t)(n0 + (idx >> 2)) * K + k0 + (idx & 3) * 8;
        rb[S][i] = *reinterpret_cast<const f32x4*>(Ch + at);
        rb[S][2 + i] = *reinterpret_cast<const f32x4*>(Cl + at);
      }
    }
  };
  auto split_store = [&](const f32x4& x, int row, __bf16* hi_row, __bf16* lo_row) {
    bf16x4 h, l;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      h[u] = (__bf16)x[u];
      l[u] = (__bf16)(x[u] - (float)h[u]);
    }
    const int at = x3_chunk(row, sc4 >> 1) + (sc4 & 1) * 4;
    *reinterpret_cast<bf16x4*>(hi_row + at) = h;
    *reinterpret_cast<bf16x4*>(lo_row + at) = l;
  };
  auto lstore = [&](auto stag, int buf) {
    constexpr int S = decltype(stag)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = srow + 32 * i;
      if constexpr (!(PRE & 1)) split_store(ra[S][i], row, Tp(buf, 0, 0) + row * kX3Pitch, Tp(buf, 0, 1) + row * kX3Pitch);
      if constexpr (!(PRE & 2)) split_store(rb[S][i], row, Tp(buf, 1, 0) + row * kX3Pitch, Tp(buf, 1, 1) + row * kX3Pitch);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i, row = idx >> 2, at = row * kX3Pitch + x3_chunk(row, idx & 3);
      if constexpr (PRE & 1) {
        *reinterpret_cast<f32x4*>(Tp(buf, 0, 0) + at) = ra[S][i];
        *reinterpret_cast<f32x4*>(Tp(buf, 0, 1) + at) = ra[S][2 + i];
      }
      if constexpr (PRE & 2) {
        *reinterpret_cast<f32x4*>(Tp(buf, 1, 0) + at) = rb[S][i];
        *reinterpret_cast<f32x4*>(Tp(buf, 1, 1) + at) = rb[S][2 + i];
      }
    }
  };
  const int r = lane & 31, hh = lane >> 5;
  auto compute = [&](int buf) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {  // two k-steps of 16
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ko = x3_chunk(r, 2 * s2 + hh);  // (the row offsets wr * 64 + t * 32 do not change (row >> 2) & 3)
        ah[t] = *reinterpret_cast<const bf16x8*>(Tp(buf, 0, 0) + (wr * 64 + t * 32 + r) * kX3Pitch + ko);
        al[t] = *reinterpret_cast<const bf16x8*>(Tp(buf, 0, 1) + (wr * 64 + t * 32 + r) * kX3Pitch + ko);
        bh[t] = *reinterpret_cast<const bf16x8*>(Tp(buf, 1, 0) + (wc * 64 + t * 32 + r) * kX3Pitch + ko);
        bl[t] = *reinterpret_cast<const bf16x8*>(Tp(buf, 1, 1) + (wc * 64 + t * 32 + r) * kX3Pitch + ko);
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);  // small terms first
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
        }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  // tiles: t = k0 / 32.  Registers hold tiles t+1 (slot (t+1)&1) and t+2 while LDS buffer t&1 is read.  K is a multiple of
  // 64 (kColAlign): an even number of tiles, so the loop body is a straight pair of steps; past the end the prefetch re-reads
  // the last tile and the store goes to a buffer nobody reads any more (branches around them cost the compiler's counted
  // waits and a copy of the 64 accumulator registers per step).
  const uint32_t k_tiles = K / kGemmBK, k_last = K - kGemmBK;
  auto kclamp = [&](uint32_t t) { const uint32_t k0 = t * kGemmBK; return k0 < k_last ? k0 : k_last; };
  gload(S0{}, 0);
  gload(S1{}, kclamp(1));
  lstore(S0{}, 0);
  gload(S0{}, kclamp(2));
  __syncthreads();
  for (uint32_t t = 0; t < k_tiles; t += 2) {
    // even tile t: LDS buffer 0; registers: slot 1 = tile t+1, slot 0 = tile t+2
    lstore(S1{}, 1);             // tile t+1 -> buffer 1 (its readers finished before the last barrier)
    gload(S1{}, kclamp(t + 3));
    compute(0);
    __syncthreads();
    // odd tile t+1: LDS buffer 1; registers: slot 0 = tile t+2, slot 1 = tile t+3
    lstore(S0{}, 0);
    gload(S0{}, kclamp(t + 4));
    compute(1);
    __syncthreads();
  }
  __syncthreads();  // (the epilogue re-uses the operand storage)
  gemm_epilogue<NORM_ROWS>(acc, reinterpret_cast<float*>(T), cnorm, N_pad, G, metric, k_rows, part_v1, part_c1, part_v2, m0, n0, wr, wc, r, hh);
}

// ---- the assign contraction on 256 x 256 block tiles ---------------------------------------------------------------------
// The 128 x 128 kernel above moves 32 KB of operands from L2 / Infinity Cache into LDS per 3.1 MFLOP of bf16 products; with
// two blocks per CU on 256 CUs that is 11.5 TB/s at its measured 305 algorithmic TFLOP/s -- the wall it runs into (a split of
// the point operand hoisted out of the kernel changed nothing: the VALU work hides behind that traffic; MFMA-busy 44 %).
// Here a block of EIGHT waves owns 256 centroids x 256 points: each wave a 64 x 128 sub-tile (2 x 4 MFMA tiles, 128
// accumulator registers), 64 KB of operands per K-tile for FOUR times the products -- half the bytes per flop from L2, and 12
// instead of 16 fragment reads per 24 MFMAs from LDS.  One block per CU (128 KB of LDS: two buffers), two waves per SIMD as
// before.  Assign pass only (NORM_ROWS: M = centroids, pre-split into bf16 hi | lo once per pass; N = the point batch, f32,
// split while it is staged); k_pad and the batch must be multiples of 256, otherwise the 128 x 128 kernel runs.
// The coarse quantiser (1024 x 4096) would be 64 such blocks on 256 CUs and stays on the 128 x 128 tiles.
constexpr int kGemmWide = 256;
constexpr size_t kX3WLdsBytes = 2 * 2 * 2 * (size_t)kGemmWide * kX3Pitch * 2;  // two buffers x A|B x hi|lo = 128 KB
// Epilogue of the wide (256 x 256) contraction kernels: per point (column) the smallest value, its centroid and the second smallest over each
// 128-centroid row tile (the unit assign_argmin_merge_kernel / assign_tile_rescan_kernel work in): same rules as gemm_epilogue<true>.
// `T`: the block's operand storage, dead by now (the callers' loops end on a barrier with no load to LDS in flight).
// CN_STAGED: the caller has put the staged norms at T + 2048 floats (and passed a barrier) -- and may have LDS-DMA loads in flight: the
// barriers here wait for LDS traffic only (a __syncthreads() would drain the loads).
template <bool CN_STAGED = false>
__device__ __forceinline__ void wide_epilogue(f32x16 (&acc)[2][4], void* T, const float* __restrict__ cnorm, uint32_t N_pad, int metric, uint32_t k_rows,
                                              float* __restrict__ part_v1, uint32_t* __restrict__ part_c1, float* __restrict__ part_v2, uint32_t m0,
                                              uint32_t n0, int wr, int wc, int r, int hh) {
  auto fold = [](float& v1, uint32_t& c1, float& v2, float w1, uint32_t d1, float w2) {
    const bool nan = (v2 != v2) || (w2 != w2);
    if (w1 < v1) { v2 = v1 < w2 ? v1 : w2; v1 = w1; c1 = d1; }
    else { const float t = w1 < v2 ? w1 : v2; v2 = t; }
    if (nan) v2 = __builtin_nanf("");
  };
  // the block's 256 centroid norms through LDS, four consecutive rows per ds_read_b128.  (Through round 6's first half every element read
  // cnorm[m] from memory, and with no register to spare the compiler waited for each of the 128 loads of a lane before it issued the
  // next: 15 us of a block's 42 -- 500 us of the 1.29 ms launch at k = 4096 x 131072 points.)
  // Staged per metric so that an element is  g = cn - s * acc  with no further case: squared Euclidean s = 2, cn = |c|^2; cosine distance
  // s = 1, cn = 0 (0 - acc: -acc but for the sign of a zero, which no comparison sees); padding centroids (zero rows) cn = +inf: never a
  // candidate.
  float* const cn_s = reinterpret_cast<float*>(T) + 2048;  // (behind the exchange area below: 1536 floats)
  if constexpr (!CN_STAGED) {
    if (threadIdx.x < (unsigned)kGemmWide) {
      const uint32_t m = m0 + threadIdx.x;
      cn_s[threadIdx.x] = m >= k_rows ? __builtin_inff() : (metric ? 0.0f : cnorm[m]);
    }
    __syncthreads();
  }
  f32x4 cn4[2][4];
  bool cn_nan = false;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      cn4[a][q] = *reinterpret_cast<const f32x4*>(cn_s + wr * 64 + a * 32 + 8 * q + 4 * hh);
#pragma unroll
      for (int u = 0; u < 4; ++u) cn_nan |= cn4[a][q][u] != cn4[a][q][u];
    }
  const float sc = metric ? 1.0f : 2.0f;
  float bv1[4], bv2[4];
  uint32_t bc1[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    // Seven operations per element (65,536 elements per block: the epilogue is VALU time -- a fifth of a tile's at k = 65536).  A tie with the candidate makes second == best, a
    // NaN makes second NaN (below): neither certifies.  A non-finite accumulator -- the only way, besides a NaN norm, an element becomes NaN
    // -- turns `z` NaN through z = fma(acc, 0, z) (an infinite accumulator too: it would not certify anything worth having).
    float v1 = __builtin_inff(), v2 = __builtin_inff();
    f32x2_t z2 = {0.0f, 0.0f};
    uint32_t c1 = m0 + wr * 64 + 4 * hh;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int e2 = 0; e2 < 8; ++e2) {
        // two elements per packed operation where there is one (v_pk_fma_f32): g = cn - s acc as fma(-s, acc, cn) -- s acc is exact
        // (s = 1 or 2), so the fused result is the subtraction's, bit for bit -- and the non-finite detector
        const f32x2_t av = {acc[a][b][2 * e2], acc[a][b][2 * e2 + 1]};
        const f32x2_t cn = {cn4[a][e2 >> 1][(2 * e2) & 3], cn4[a][e2 >> 1][(2 * e2 + 1) & 3]};
        z2 = __builtin_elementwise_fma(av, f32x2_t{0.0f, 0.0f}, z2);
        const f32x2_t g2 = __builtin_elementwise_fma(f32x2_t{-sc, -sc}, av, cn);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = 2 * e2 + u;
          const uint32_t m = m0 + wr * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          const float g = g2[u];
          const bool lt = g < v1;
          const float mx = lt ? v1 : g;   // the larger of (best so far, g); g when they are equal
          c1 = lt ? m : c1;
          v2 = mx < v2 ? mx : v2;
          v1 = lt ? g : v1;
        }
      }
    const float z = z2[0] + z2[1];
    if (cn_nan || z != z) v2 = __builtin_nanf("");
    const float w1 = __shfl_xor(v1, 32, kWave), w2 = __shfl_xor(v2, 32, kWave);  // the other half of the rows sits in lane ^ 32
    const uint32_t d1 = (uint32_t)__shfl_xor((int)c1, 32, kWave);
    fold(v1, c1, v2, w1, d1, w2);
    bv1[b] = v1; bc1[b] = c1; bv2[b] = v2;
  }
  // rows 64..127 of a 128-centroid tile belong to the odd wave rows: through LDS (the operand tiles are dead by now: the
  // loop ended on a barrier)
  float* xs = reinterpret_cast<float*>(T);  // [wr >> 1][wc][b][r][3]: 1536 floats
  if ((wr & 1) == 1 && hh == 0) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float* t = xs + ((((wr >> 1) * 2 + wc) * 4 + b) * 32 + r) * 3;
      t[0] = bv1[b]; t[1] = __uint_as_float(bc1[b]); t[2] = bv2[b];
    }
  }
  if constexpr (CN_STAGED) lds_barrier(); else __syncthreads();
  if ((wr & 1) == 0 && hh == 0) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float* t = xs + ((((wr >> 1) * 2 + wc) * 4 + b) * 32 + r) * 3;
      fold(bv1[b], bc1[b], bv2[b], t[0], __float_as_uint(t[1]), t[2]);
      const uint64_t o = (uint64_t)(m0 / kGemmBM + (wr >> 1)) * N_pad + n0 + wc * 128 + b * 32 + r;
      part_v1[o] = bv1[b]; part_c1[o] = bc1[b]; part_v2[o] = bv2[b];
    }
  }
}
// The tile step:  fragment reads -> [MFMAs with the SPLIT of the next tile's point operand between them] x 2 k-steps, with
// scheduling groups that ask for one MFMA followed by two VALU instructions -- the split's conversions issue in the shadow of
// the matrix cores (a wave cannot issue its next MFMA for ~28 cycles anyway) instead of in a phase of their own in which, the
// eight waves of the block moving in lockstep, the matrix cores idle -- and the next-but-one tile's pieces requested as soon as
// their staging registers are free.  (The template parameter is what is left of rounds 3-4's schedule variants -- plain order,
// interleaved, both operands pre-split by LDS-DMA / through registers: DESIGN.md Appendix A and the history of this file; the
// instantiation keeps the kernel's name in the committed profiles.)
// TERMS = 1 (round 6): ONLY the <hi, hi> product -- a third of the MFMAs, half the LDS traffic and split work -- as the FIRST filter of
// a cascade: its values err by up to 2^-8 (|x|^2 + |c|^2) (kX1Slack), the certificate of assign_rescore_kernel is that much wider, and
// the points it cannot settle go to the tile-limited exact re-scan (assign_tile_rescan_kernel) instead of a second contraction.  On
// clustered data almost every point's nearest centroid beats the runner-up by far more than that (km_assign_mfma decides per pass
// from its first batch).
template <int SCHED, int TERMS = 3>  // TERMS = 1: Ch holds FP16 bit patterns, the points are converted to fp16
static __global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void dist_gemm_x3w_kernel(
    const float* __restrict__ X, const __bf16* __restrict__ Ch, const __bf16* __restrict__ Cl, const float* __restrict__ cnorm, uint32_t K,
    uint32_t N_pad, int metric, uint32_t m_tiles, uint32_t n_tiles, uint32_t grp, uint32_t k_rows, float* __restrict__ part_v1,
    uint32_t* __restrict__ part_c1, float* __restrict__ part_v2) {
  static_assert(SCHED == 2, "one schedule is shipped");
  static_assert(TERMS == 3 || TERMS == 1, "hi*hi + hi*lo + lo*hi, or hi*hi alone");
  constexpr bool LO = TERMS == 3;
  extern __shared__ __attribute__((aligned(16))) __bf16 T[];
  constexpr int kPart = kGemmWide * kX3Pitch;
  auto Tp = [&](int buf, int mat, int part) { return T + ((buf * 2 + mat) * 2 + part) * kPart; };
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;  // wave rows 0..3 (64 centroids each), wave columns 0..1 (128 points each)
  uint32_t tile_m, tile_n;
  gemm_tile_coords(m_tiles, n_tiles, grp, tile_m, tile_n);
  const uint32_t m0 = tile_m * kGemmWide, n0 = tile_n * kGemmWide;
  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;
  // staging registers of ONE tile: centroids 256 rows x 4 chunks of 16 bytes per part (thread t: chunks t, t + 512), points
  // 256 rows x 8 float4 (thread t: slots t + 512 i, i < 4)
  f32x4 ra[4], rb[4];
  auto gload = [&](uint32_t k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i;
      const uint64_t at = (uint64_t)(m0 + (idx >> 2)) * K + k0 + (idx & 3) * 8;
      ra[i] = *reinterpret_cast<const f32x4*>(Ch + at);
      if constexpr (LO) ra[2 + i] = *reinterpret_cast<const f32x4*>(Cl + at);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 512 * i;
      rb[i] = *reinterpret_cast<const f32x4*>(X + (uint64_t)(n0 + (idx >> 3)) * K + k0 + (idx & 7) * 4);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 512 * i, row = idx >> 2, at = row * kX3Pitch + x3_chunk(row, idx & 3);
      *reinterpret_cast<f32x4*>(Tp(buf, 0, 0) + at) = ra[i];
      if constexpr (LO) *reinterpret_cast<f32x4*>(Tp(buf, 0, 1) + at) = ra[2 + i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + 512 * i, row = idx >> 3, c4 = idx & 7;
      const int at = row * kX3Pitch + x3_chunk(row, c4 >> 1) + (c4 & 1) * 4;
      if constexpr (LO) {
        bf16x4 h, l;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          h[u] = (__bf16)rb[i][u];
          l[u] = (__bf16)(rb[i][u] - (float)h[u]);
        }
        *reinterpret_cast<bf16x4*>(Tp(buf, 1, 0) + at) = h;
        *reinterpret_cast<bf16x4*>(Tp(buf, 1, 1) + at) = l;
      } else {
        gf16x4 h;
#pragma unroll
        for (int u = 0; u < 4; ++u) h[u] = (_Float16)rb[i][u];
        *reinterpret_cast<gf16x4*>(Tp(buf, 1, 0) + at) = h;
      }
    }
  };
  const int r = lane & 31, hh = lane >> 5;
  // tile t is computed out of LDS buffer t & 1 while the registers bring tile t + 2 in; past the end the prefetch re-reads the
  // last tile and the store goes to a buffer nobody reads any more (no branches around loads: the compiler's counted waits)
  const uint32_t k_tiles = K / kGemmBK, k_last = K - kGemmBK;
  auto kclamp = [&](uint32_t t) { const uint32_t k0 = t * kGemmBK; return k0 < k_last ? k0 : k_last; };
  gload(0);
  lstore(0);
  gload(kclamp(1));
  __syncthreads();
  {
    // The global loads of tile t + 2 are issued as soon as tile t + 1's registers are consumed instead of at the end of the step:
    // there the loads went out just before the barrier and their first consumer -- the split at the top of the next step -- sat
    // just behind it, so every K-tile began with the whole block waiting out an L2 / Infinity-Cache round trip
    // (PMC, round 4: a K-tile takes 5.9 k cycles of which the matrix cores are busy 3.1 k).  Here the centroid pieces of tile
    // t + 1 go to LDS at the TOP of step t (the other buffer's readers left at the last barrier) and their registers take tile
    // t + 2 at once; each half of the point slots is reloaded right behind its split.  No register is added: round 3's try at
    // loading a tile earlier kept a third tile in flight and spilled.
    auto gload_a = [&](uint32_t k0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int idx = tid + 512 * i;
        const uint64_t at = (uint64_t)(m0 + (idx >> 2)) * K + k0 + (idx & 3) * 8;
        ra[i] = *reinterpret_cast<const f32x4*>(Ch + at);
        if constexpr (LO) ra[2 + i] = *reinterpret_cast<const f32x4*>(Cl + at);
      }
    };
    auto gload_b = [&](int i, uint32_t k0) {
      const int idx = tid + 512 * i;
      rb[i] = *reinterpret_cast<const f32x4*>(X + (uint64_t)(n0 + (idx >> 3)) * K + k0 + (idx & 7) * 4);
    };
    bf16x8 ah[2], al[2], bh[4], bl[4];
    for (uint32_t t = 0; t < k_tiles; ++t) {
      const int buf = (int)(t & 1), nbuf = buf ^ 1;
      const uint32_t k2 = kclamp(t + 2);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int ko = x3_chunk(r, 2 * s2 + hh);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          ah[tt] = *reinterpret_cast<const bf16x8*>(Tp(buf, 0, 0) + (wr * 64 + tt * 32 + r) * kX3Pitch + ko);
          if constexpr (LO) al[tt] = *reinterpret_cast<const bf16x8*>(Tp(buf, 0, 1) + (wr * 64 + tt * 32 + r) * kX3Pitch + ko);
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          bh[tt] = *reinterpret_cast<const bf16x8*>(Tp(buf, 1, 0) + (wc * 128 + tt * 32 + r) * kX3Pitch + ko);
          if constexpr (LO) bl[tt] = *reinterpret_cast<const bf16x8*>(Tp(buf, 1, 1) + (wc * 128 + tt * 32 + r) * kX3Pitch + ko);
        }
        if (s2 == 0) {  // tile t + 1's centroid pieces -> the other buffer; their registers take tile t + 2
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int idx = tid + 512 * i, row = idx >> 2, at = row * kX3Pitch + x3_chunk(row, idx & 3);
            *reinterpret_cast<f32x4*>(Tp(nbuf, 0, 0) + at) = ra[i];
            if constexpr (LO) *reinterpret_cast<f32x4*>(Tp(nbuf, 0, 1) + at) = ra[2 + i];
          }
          gload_a(k2);
        }
        // (hard scheduling fences between the phases of a k-step: left to itself the compiler hoists both halves of the split --
        // and with them the waits for their loads -- in front of the first MFMA and sinks the reloads to the end of the step)
        __builtin_amdgcn_sched_barrier(0);
        auto mfma_row = [&](int a) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            if constexpr (LO) {
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);  // small terms first
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
            } else {
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gf16x8, ah[a]), __builtin_bit_cast(gf16x8, bh[b]), acc[a][b], 0, 0, 0);
            }
          }
        };
        // first half of the k-step's MFMAs with this k-step's half of the split between them, stored at once
#pragma unroll
        for (int i = 2 * s2; i < 2 * s2 + 2; ++i) {
          const int idx = tid + 512 * i, row = idx >> 3, c4 = idx & 7;
          const int at = row * kX3Pitch + x3_chunk(row, c4 >> 1) + (c4 & 1) * 4;
          if constexpr (LO) {
            bf16x4 sh, sl;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              sh[u] = (__bf16)rb[i][u];
              sl[u] = (__bf16)(rb[i][u] - (float)sh[u]);
            }
            *reinterpret_cast<bf16x4*>(Tp(nbuf, 1, 0) + at) = sh;
            *reinterpret_cast<bf16x4*>(Tp(nbuf, 1, 1) + at) = sl;
          } else {
            gf16x4 sh;
#pragma unroll
            for (int u = 0; u < 4; ++u) sh[u] = (_Float16)rb[i][u];
            *reinterpret_cast<gf16x4*>(Tp(nbuf, 1, 0) + at) = sh;
          }
        }
        mfma_row(0);
#pragma unroll
        for (int g = 0; g < 4 * TERMS; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // two VALU
        }
        __builtin_amdgcn_sched_barrier(0);
        // the split slots' registers take tile t + 2; second half of the MFMAs
        gload_b(2 * s2, k2);
        gload_b(2 * s2 + 1, k2);
        mfma_row(1);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  }
  wide_epilogue(acc, T, cnorm, N_pad, metric, k_rows, part_v1, part_c1, part_v2, m0, n0, wr, wc, r, hh);
}
// The cascade's first filter with BOTH operands already fp16 in memory (round 6, second half): the same 256 x 256 block, 8 waves of
// 64 x 128, the same products in the same order as dist_gemm_x3w_kernel<2, 1> -- bit for bit the same values -- but K-tiles of 64 columns
// staged by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, no staging registers, no conversion in the loop) in pieces of
// 8 rows x one whole 128-byte line.  What bounds the register-staged kernel is not the matrix cores (busy 26 % of a K-tile's 4.2 k cycles,
// profiles/r06_kmeans.json) but the CU's fill path: its 32-column K-tiles take HALF a cache line of every fp16 row per step, the other
// half is gone from the 32 KB L1 when the next step asks for it (512 lines per step), so a K-tile pulls 64 KB through a 64 B/clk port
// for 32 KB of operands.  A ring of four such half-line tiles three steps ahead measured the same 1.36 ms per launch at k = 4096
// (and 17.6 vs 20.4 ms at k = 65536): depth was not it.  Whole lines halve the fill traffic and the address work per flop.
//   LDS image: rows of 128 bytes (64 columns), the 16-byte chunk c of row `row` at position c ^ ((row >> 1) & 7): the 16 lanes a
//   ds_read_b128 serves per clock (rows r .. r + 15, one k-chunk) fall on the 16 slots of the 256-byte bank row.  An LDS-DMA writes lane
//   L's 16 bytes at base + 16 L, so the swizzle sits on the SOURCE side: lane L of the piece that covers rows 8 j .. 8 j + 7 fetches
//   chunk (L & 7) ^ ((row >> 1) & 7) of row 8 j + (L >> 3) -- the eight lanes of a row still read one whole line.
//   Ordering (LDS-DMA data is ordered for a ds_read only by the issuing wave's vmcnt followed by a barrier the reader has passed):
//   step t issues tile t + 1 into the other buffer (read in step t - 1: every wave left it at the barrier that ended that step, its
//   reads returned -- lgkmcnt(0) in front of the barrier), multiplies tile t, waits for its own pieces and meets the others.
//   Persistent: a block per CU walks the tiles L = blockIdx.x, + gridDim.x, ... (the XCD of tile L is L & 7 either way: gemm_tile_coords).
//   The last step of a tile issues the FIRST K-tile of the block's next tile instead of nothing and does not wait for it: it lands
//   while the epilogue runs (whose scratch -- exchange area and staged norms -- sits behind the two buffers, and whose barriers wait for
//   LDS traffic only).  A block per tile paid the first tile's round trip (2.5 us) and the epilogue's VALU time (4 us) 32 times per CU
//   with nothing else on the CU to run under them.
constexpr int kHBK = 64;                        // columns of a K-tile: one 128-byte line of fp16
constexpr int kHTile = kGemmWide * kHBK;        // elements of one operand's K-tile (32 KB)
constexpr size_t kGemmHLdsBytes = (size_t)2 * 2 * kHTile * 2 + 10240;  // two buffers x A | B = 128 KB, + the epilogue's scratch
static __global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void dist_gemm_h_kernel(
    const _Float16* __restrict__ Xh, const _Float16* __restrict__ Ch, const float* __restrict__ cnorm, uint32_t K, uint32_t N_pad, int metric,
    uint32_t m_tiles, uint32_t n_tiles, uint32_t grp, uint32_t k_rows, float* __restrict__ part_v1, uint32_t* __restrict__ part_c1,
    float* __restrict__ part_v2) {
  extern __shared__ __attribute__((aligned(16))) _Float16 TH[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  float* const scratch = reinterpret_cast<float*>(TH + 4 * kHTile);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;  // wave rows 0..3 (64 centroids each), wave columns 0..1 (128 points each)
  const int r = lane & 31, hh = lane >> 5;
  const int rsw = (r >> 1) & 7;
  const uint32_t k_tiles = K / kHBK, total = m_tiles * n_tiles;
  // this wave's eight pieces of a K-tile: rows 8 (wid + 8 i) .. + 7 of the centroid tile and of the point tile, i < 4
  // ((row >> 1) & 7 of row 8 j + (L >> 3), j = wid + 8 i: 4 (wid & 1) + (L >> 4))
  const int prow = lane >> 3, pchunk = (lane & 7) ^ (4 * (wid & 1) + (lane >> 4));
  // (byte offsets in 32 bits from the operands' bases -- wave-uniform, in scalar registers: both operands are below 4 GB; launch_gemm_h checks)
  const uint32_t piece_step = 64u * K * 2u;
  auto tile_at = [&](uint32_t L, uint32_t& m0, uint32_t& n0, uint32_t& aoff, uint32_t& boff) {
    uint32_t tile_m, tile_n;
    gemm_tile_coords(m_tiles, n_tiles, grp, tile_m, tile_n, L);
    m0 = tile_m * kGemmWide; n0 = tile_n * kGemmWide;
    aoff = ((m0 + 8u * (uint32_t)wid + (uint32_t)prow) * K + (uint32_t)pchunk * 8u) * 2u;   // piece i: + 64 i rows
    boff = ((n0 + 8u * (uint32_t)wid + (uint32_t)prow) * K + (uint32_t)pchunk * 8u) * 2u;
  };
  const char* const Cb = reinterpret_cast<const char*>(Ch);
  const char* const Xb = reinterpret_cast<const char*>(Xh);
  auto issue = [&](uint32_t aoff, uint32_t boff, uint32_t k0, int buf) {
    _Float16* const A = TH + (buf * 2 + 0) * kHTile + 8 * wid * kHBK;
    _Float16* const B = TH + (buf * 2 + 1) * kHTile + 8 * wid * kHBK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((glb_void*)(Cb + (aoff + (uint32_t)i * piece_step + k0 * 2u)), (lds_void*)(A + i * 64 * kHBK), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void*)(Xb + (boff + (uint32_t)i * piece_step + k0 * 2u)), (lds_void*)(B + i * 64 * kHBK), 16, 0, 0);
    }
  };
  uint32_t L = blockIdx.x;
  if (L >= total) return;
  uint32_t m0, n0, asrc, bsrc;
  tile_at(L, m0, n0, asrc, bsrc);
  issue(asrc, bsrc, 0, 0);
  for (;;) {
    const uint32_t Ln = L + gridDim.x;
    const bool more = Ln < total;  // (block-uniform)
    uint32_t m0n = 0, n0n = 0, asrc_n = asrc, bsrc_n = bsrc;
    if (more) tile_at(Ln, m0n, n0n, asrc_n, bsrc_n);
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;
    if (tid < kGemmWide) {  // the tile's norms as the epilogue wants them (wide_epilogue<true>)
      const uint32_t m = m0 + (uint32_t)tid;
      scratch[2048 + tid] = m >= k_rows ? __builtin_inff() : (metric ? 0.0f : cnorm[m]);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the tile's first K-tile has landed (issued a tile ago)
    __builtin_amdgcn_sched_barrier(0);
    auto step = [&](auto btag, auto ltag, uint32_t t) {
      constexpr int BUF = decltype(btag)::value;
      constexpr bool LAST = decltype(ltag)::value;
      if constexpr (!LAST) issue(asrc, bsrc, (t + 1) * kHBK, BUF ^ 1);
      else if (more) issue(asrc_n, bsrc_n, 0, BUF ^ 1);
      const _Float16* const A = TH + (BUF * 2 + 0) * kHTile;
      const _Float16* const B = TH + (BUF * 2 + 1) * kHTile;
      // the fragments of k-step s + 1 are requested BEFORE the MFMAs of k-step s (two register sets): all eight waves run in step --
      // what the barrier leaves of it -- so reads, then MFMAs, would keep the LDS port (768 cycles of reads per 32 columns) and the
      // matrix cores (1024) busy one after the other
      gf16x8 fa[2][2], fb[2][4];
      auto fetch = [&](auto stag, int s4) {
        constexpr int S = decltype(stag)::value;
        const int ko = ((2 * s4 + hh) ^ rsw) * 8;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) fa[S][tt] = *reinterpret_cast<const gf16x8*>(A + (wr * 64 + tt * 32 + r) * kHBK + ko);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) fb[S][tt] = *reinterpret_cast<const gf16x8*>(B + (wc * 128 + tt * 32 + r) * kHBK + ko);
      };
      auto mult = [&](auto stag) {
        constexpr int S = decltype(stag)::value;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S][a], fb[S][b], acc[a][b], 0, 0, 0);
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      fetch(I0{}, 0);
      fetch(I1{}, 1);
      __builtin_amdgcn_sched_barrier(0);
      mult(I0{});
      __builtin_amdgcn_sched_barrier(0);
      fetch(I0{}, 2);
      __builtin_amdgcn_sched_barrier(0);
      mult(I1{});
      __builtin_amdgcn_sched_barrier(0);
      fetch(I1{}, 3);
      __builtin_amdgcn_sched_barrier(0);
      mult(I0{});
      __builtin_amdgcn_sched_barrier(0);
      mult(I1{});
      // (lgkmcnt: this wave's reads of the buffer have returned before anybody re-fills it; LAST: the next tile's pieces stay in flight)
      if constexpr (LAST) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    uint32_t t = 0;
    for (; t + 2 < k_tiles; t += 2) {  // (an even number of K-tiles: gemm_h_ok)
      step(B0{}, std::false_type{}, t);
      step(B1{}, std::false_type{}, t + 1);
    }
    step(B0{}, std::false_type{}, t);
    step(B1{}, std::true_type{}, t + 1);
    wide_epilogue<true>(acc, scratch, cnorm, N_pad, metric, k_rows, part_v1, part_c1, part_v2, m0, n0, wr, wc, r, hh);
    if (!more) break;
    L = Ln; m0 = m0n; n0 = n0n; asrc = asrc_n; bsrc = bsrc_n;
  }
}
// rows of a point batch -> fp16 (round to nearest: what dist_gemm_x3w_kernel<2, 1> converts on the fly), columns d .. ld_out and rows
// n_rows .. n_pad zero: thread per 8 columns
static __global__ __launch_bounds__(256) void rows_to_f16_pad_kernel(const float* __restrict__ x, uint64_t ldx, uint32_t d, uint32_t n_rows, uint32_t n_pad,
                                                                     uint32_t ld_out, _Float16* __restrict__ out) {
  const uint32_t per_row = ld_out / 8;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)n_pad * per_row) return;
  const uint32_t row = (uint32_t)(i / per_row), c0 = (uint32_t)(i % per_row) * 8;
  gf16x8 h;
  if (row < n_rows && c0 + 8 <= d && (ldx & 3u) == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0) {  // (the common case: two 16-byte loads)
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + (uint64_t)row * ldx + c0);
    const f32x4 b = *reinterpret_cast<const f32x4*>(x + (uint64_t)row * ldx + c0 + 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) { h[u] = (_Float16)a[u]; h[4 + u] = (_Float16)b[u]; }
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const uint32_t c = c0 + (uint32_t)u;
      h[u] = (row < n_rows && c < d) ? (_Float16)x[(uint64_t)row * ldx + c] : (_Float16)0.0f;
    }
  }
  *reinterpret_cast<gf16x8*>(out + (uint64_t)row * ld_out + c0) = h;
}
inline bool gemm_h_ok(uint32_t K) { return K % (2 * kHBK) == 0; }  // (+ both operands below 4 GB: km_assign_mfma's batches and k_pad x K x 2 are)
inline hipError_t launch_gemm_h(uint32_t k_pad, uint32_t nb_pad, hipStream_t st, const _Float16* Xh, const _Float16* Ch, const float* cnorm, uint32_t K,
                                uint32_t N_pad, uint32_t metric, uint32_t k_rows, float* part_v1, uint32_t* part_c1, float* part_v2) {
  static const hipError_t attr = hipFuncSetAttribute((const void*)dist_gemm_h_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmHLdsBytes);
  if (attr != hipSuccess) return attr;
  const uint32_t m_tiles = k_pad / kGemmWide, n_tiles = nb_pad / kGemmWide;
  uint32_t grp = 0;
  // (the 32 tiles an XCD has in flight: grp centroid tiles x 32 / grp point tiles.  k = 65536: 12.36 ms with 2, 12.13 with 4, 12.19 with 8)
  for (uint32_t g : {4u, 2u, 1u}) if (m_tiles % (8u * g) == 0) { grp = g; break; }
  // one block per CU (138 KB of LDS each), persistent
  static int n_cu_of[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
  if (n_cu_of[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    n_cu_of[dev] = n;
  }
  const uint32_t total = m_tiles * n_tiles;
  uint32_t grid = (uint32_t)n_cu_of[dev] / 8u * 8u;  // (a multiple of the 8 XCDs: tile L stays on XCD L & 7)
  if (grid == 0 || grid > total) grid = total;
  hipLaunchKernelGGL(dist_gemm_h_kernel, dim3(grid), dim3(512), kGemmHLdsBytes, st, Xh, Ch, cnorm, K, N_pad, (int)metric, m_tiles, n_tiles, grp, k_rows,
                     part_v1, part_c1, part_v2);
  return hipGetLastError();
}
// assign pass through the wide kernel whenever the shapes allow it
inline bool gemm_wide_ok(uint32_t k_pad, uint32_t nb_pad, bool have_split) {
  return have_split && k_pad % kGemmWide == 0 && nb_pad % kGemmWide == 0;
}
inline hipError_t launch_gemm_wide(uint32_t k_pad, uint32_t nb_pad, hipStream_t st, const float* X, const __bf16* ch, const __bf16* cl, const float* cnorm,
                                   uint32_t K, uint32_t N_pad, uint32_t metric, uint32_t k_rows, float* part_v1, uint32_t* part_c1, float* part_v2,
                                   bool hi_only = false) {
  static const hipError_t attr = hipFuncSetAttribute((const void*)dist_gemm_x3w_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kX3WLdsBytes);
  if (attr != hipSuccess) return attr;
  static const hipError_t attr1 = hipFuncSetAttribute((const void*)dist_gemm_x3w_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kX3WLdsBytes);
  if (attr1 != hipSuccess) return attr1;
  const uint32_t m_tiles = k_pad / kGemmWide, n_tiles = nb_pad / kGemmWide;
  // each XCD keeps `grp` centroid tiles (256 rows x K x 4 B of hi | lo = 768 KB at K = 768) in its L2 and walks the point tiles
  uint32_t grp = 0;
  for (uint32_t g : {2u, 1u}) if (m_tiles % (8u * g) == 0) { grp = g; break; }
  if (hi_only)
    hipLaunchKernelGGL((dist_gemm_x3w_kernel<2, 1>), dim3(m_tiles * n_tiles), dim3(512), kX3WLdsBytes, st, X, ch, cl, cnorm, K, N_pad, (int)metric, m_tiles,
                       n_tiles, grp, k_rows, part_v1, part_c1, part_v2);
  else
    hipLaunchKernelGGL(dist_gemm_x3w_kernel<2>, dim3(m_tiles * n_tiles), dim3(512), kX3WLdsBytes, st, X, ch, cl, cnorm, K, N_pad, (int)metric, m_tiles,
                       n_tiles, grp, k_rows, part_v1, part_c1, part_v2);
  return hipGetLastError();
}

// Launch of either contraction kernel (the bf16x3 one needs the attribute for its 80 KB of dynamic LDS: set once per
// instantiation and process).
// sh / sl: the centroid operand pre-split by split_bf16_kernel (nullable: split on the fly) -- the M operand of the assign
// pass (NORM_ROWS), the N operand of the coarse quantiser.
template <bool NORM_ROWS>
inline hipError_t launch_gemm(bool x3, uint32_t m_tiles, uint32_t n_tiles, hipStream_t st, const float* Q, const float* C, const float* cnorm,
                              uint32_t K, uint32_t N_pad, float* G, uint32_t metric, uint32_t k_rows = 0, float* part_v1 = nullptr,
                              uint32_t* part_c1 = nullptr, float* part_v2 = nullptr, const __bf16* sh = nullptr, const __bf16* sl = nullptr,
                              uint32_t* zero = nullptr, uint32_t n_zero = 0) {  // zero / n_zero: the bf16x3 kernel only (the caller checks x3)
  const uint32_t grp = NORM_ROWS ? gemm_tile_group(m_tiles) : 0;
  if (!x3) {
    hipLaunchKernelGGL(dist_gemm_kernel<NORM_ROWS>, dim3(m_tiles * n_tiles), dim3(256), 0, st, Q, C, cnorm, K, N_pad, G, metric, m_tiles, n_tiles,
                       grp, k_rows, part_v1, part_c1, part_v2);
    return hipGetLastError();
  }
  constexpr int kPreBit = NORM_ROWS ? 1 : 2;
  auto go = [&](auto pre_tag) {
    constexpr int PRE = decltype(pre_tag)::value;
    static const hipError_t attr = hipFuncSetAttribute((const void*)dist_gemm_x3_kernel<NORM_ROWS, PRE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)kX3LdsBytes);
    if (attr != hipSuccess) return attr;
    const __bf16* qh = NORM_ROWS ? sh : nullptr; const __bf16* ql = NORM_ROWS ? sl : nullptr;
    const __bf16* ch = NORM_ROWS ? nullptr : sh; const __bf16* cl = NORM_ROWS ? nullptr : sl;
    hipLaunchKernelGGL((dist_gemm_x3_kernel<NORM_ROWS, PRE>), dim3(m_tiles * n_tiles), dim3(256), kX3LdsBytes, st, Q, C, qh, ql, ch, cl, cnorm, K,
                       N_pad, G, metric, m_tiles, n_tiles, grp, k_rows, part_v1, part_c1, part_v2, zero, n_zero);
    return hipGetLastError();
  };
  return (sh && sl) ? go(std::integral_constant<int, kPreBit>{}) : go(std::integral_constant<int, 0>{});
}
// rows of x -> fp16 (round to nearest) and the largest squared residual |row - fp16(row)|^2 over the rows (x - fp16(x) is exact in f32:
// shadow_residual_kernel's argument; the sum of squares is inflated where it is used).  Block per row.
static __global__ __launch_bounds__(256) void to_f16_resid_kernel(const float* __restrict__ x, uint32_t ld, uint32_t n_rows, _Float16* __restrict__ out,
                                                                  uint32_t* __restrict__ rmax2_bits) {
  __shared__ float part[4];
  const uint32_t row = blockIdx.x;
  if (row >= n_rows) return;
  float acc = 0.0f;
  for (uint32_t j = threadIdx.x; j < ld; j += blockDim.x) {
    const float v = x[(uint64_t)row * ld + j];
    const _Float16 h = (_Float16)v;
    out[(uint64_t)row * ld + j] = h;
    const float dl = v - (float)h;
    acc += dl * dl;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, kWave);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = part[0] + part[1] + part[2] + part[3];
    atomicMax(rmax2_bits, t == t ? __float_as_uint(t) : 0x7F800000u);  // (t >= 0: bit order == value order; NaN counts as +inf)
  }
}
inline hipError_t launch_split_bf16(const float* x, uint64_t n_floats, __bf16* hi, __bf16* lo, hipStream_t st) {  // n_floats % 4 == 0
  const uint64_t n4 = n_floats / 4;
  hipLaunchKernelGGL(split_bf16_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x, n4, hi, lo);
  return hipGetLastError();
}

// A block of TWO waves per query: select the PS smallest G of its row, re-score them exactly, sort by the exact key,
// certify, fall back to the full exact scan for this query if the certificate fails.
// C_rm: centroids row-major [k][ldc] (pad columns zero); qp: padded queries [b][ldq].
// probe[q][0..P) receives ascending (exact distance, centroid index) keys -- the exact coarse output.
// pq.b != 0: the query's plan (plan.hip.h step 1) is made right here -- lane j holds the key of probe rank j.
// Both waves hold HALF of every 4096-value chunk of the row (32 registers per lane) and half of the candidates' rows in the exact
// re-score; the bracket's counts and the compacted keys meet in LDS (block barriers: every branch around them is block-uniform,
// both waves see the same numbers); what is cheap and needed by both -- the order statistics of the lane minima, the sort of
// the 64 candidates, |q|^2 -- is computed twice rather than exchanged.  Wave 1 leaves after the re-score; wave 0 sorts,
// certifies and plans.  (Rounds 2-4: one wave per query -- the chip's 1024 SIMDs one wave each, every phase a chain of exposed
// latencies: 26 us after this round's other changes; with two waves per SIMD the phases of different queries overlap.  Several
// QUERIES per block, their waves independent, was tried in round 4 -- VERS_SELECT_WAVES -- and bought nothing.)
constexpr int kSelWaves = 2;    // waves per query
constexpr int kSelRowsW = 32;   // candidate rows a wave stages: 64 candidates dealt alternately
static __global__ __launch_bounds__(kWave * kSelWaves) void coarse_select_rescore_kernel(
    const float* G, uint32_t N_pad, uint32_t k, const float* C_rm, uint32_t ldc, const float* qp, uint32_t ldq, uint32_t d_pad,
    float cmax2, uint32_t P, uint32_t PS, uint64_t* probe, uint32_t* status, uint32_t* fallback_count, int metric,
    unsigned long long* stamps, PlanQ pq, uint32_t n_queries) {
  const uint32_t q = blockIdx.x;
  if (q >= n_queries) return;  // (whole blocks)
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long t0 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
  unsigned long long ta = t0, tb = t0;
  const float* g = G + (uint64_t)q * N_pad;
  // (1) PS smallest approximate values; key = (order bits of G, centroid index).
  // A chunk of 4096 values sits in the two waves' registers (32 per lane, ONE round trip of independent loads).  Instead of
  // offering them to a sorted list one by one (~260 serial inserts of ~150 cycles), the PS-th smallest VALUE is bracketed on the
  // order bits -- count(g <= T) over the chunk is 32 compares per lane, a wave reduction and an exchange of two words -- until
  // PS <= count <= 64; the survivors are compacted through LDS, sorted across the lanes and merged with the best of the previous
  // chunks.  Every key among the chunk's PS smallest passes the filter (ties at T included), so `sel` is exactly what the serial
  // inserts produce.
  __shared__ uint64_t s_keys[kWave];
  __shared__ uint64_t s_x[kSelWaves][kWave];  // exchange: the waves' lists on the ordered-insert path; the exact keys
  __shared__ uint32_t s_min[kSelWaves][kWave];
  __shared__ uint32_t s_cnt[2][kSelWaves];
  __shared__ __attribute__((aligned(16))) float s_prod_all[kSelWaves][2 * staged_lds_floats(kSelRowsW)];  // the re-score's staged products (two buffers per wave)
  float* const s_prod = s_prod_all[wid];
  uint64_t sel = kKeyMax;
  constexpr int kR = 32;  // registers per lane, wave and chunk
  uint32_t cnt_it = 0;    // (parity of the count exchange: a slow wave may still be reading the previous exchange's words)
  for (uint32_t n0 = 0; n0 < k; n0 += 2 * kR * kWave) {
    uint32_t gb[kR];
    // register r of lane l of wave w holds G[n0 + 2048 w + 256 (r / 4) + 4 l + r % 4]: eight 16-byte loads per lane (any fixed
    // mapping serves: the keys carry their index).  UNCONDITIONAL loads (clamped to the row's last vector): a branch around a load
    // makes the compiler wait for each one with vmcnt(0) -- serial round trips
    auto idx_of = [&](int r) { return n0 + (uint32_t)wid * (uint32_t)(kR * kWave) + (uint32_t)(r >> 2) * 256u + 4u * (uint32_t)lane + (uint32_t)(r & 3); };
    {
      f32x4 gv[kR / 4];
#pragma unroll
      for (int r4 = 0; r4 < kR / 4; ++r4) {
        const uint32_t n = idx_of(4 * r4);
        gv[r4] = *reinterpret_cast<const f32x4*>(g + (n + 4u <= N_pad ? n : N_pad - 4u));
      }
#pragma unroll
      for (int r = 0; r < kR; ++r) gb[r] = idx_of(r) < k ? f32_to_order_bits(gv[r >> 2][r & 3]) : 0xFFFFFFFFu;
    }
    const uint32_t n_chunk = k - n0 < (uint32_t)(2 * kR * kWave) ? k - n0 : (uint32_t)(2 * kR * kWave);
    // entries of the chunk (both waves) passing `pred`; c0: wave 0's share -- where wave 1's compacted keys start
    auto count_if = [&](auto pred, uint32_t& c0) {
      uint32_t c = 0;
#pragma unroll
      for (int r = 0; r < kR; ++r) c += pred(r) ? 1u : 0u;
      c = wave_sum_u32(c);
      const uint32_t par = cnt_it & 1u;
      if (lane == 0) s_cnt[par][wid] = c;
      __syncthreads();
      c0 = s_cnt[par][0];
      const uint32_t c1 = s_cnt[par][1];
      ++cnt_it;
      return c0 + c1;
    };
    auto count_le = [&](uint32_t T, uint32_t& c0) {  // (padding entries are 0xFFFFFFFF: never counted below that)
      return count_if([&](int r) { return gb[r] <= T; }, c0);
    };
    uint32_t T = 0xFFFFFFFFu;  // a chunk of at most 64 values: all of them (padding is excluded by its index)
    uint32_t cT = 0, c0T = 0;  // entries the compaction will keep; of them in wave 0
    uint64_t cur = kKeyMax;
    bool serial = false;
    if (stamps && n0 == 0) ta = __builtin_amdgcn_s_memtime();
    if (n_chunk > (uint32_t)kWave) {
      // The bracket starts from the LANE MINIMA (lane l's = the smaller of the two waves' lane-l minima: 64 values each): PS lanes
      // hold a value at or below the PS-th smallest lane minimum, so that minimum bounds the chunk's PS-th smallest value from
      // above -- and closely: ~90 values lie at or below it.  (Rounds 2-4 interpolated between the chunk's smallest and largest
      // value: a query's few NEAR centroids lie far below the bulk, the split point crept up by a sixteenth of the range per
      // counting pass -- ~20 passes, 16 k of the kernel's 120 k cycles per query.)
      uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
      for (int r = 0; r < kR; ++r) mn = gb[r] < mn ? gb[r] : mn;
      s_min[wid][lane] = mn;
      __syncthreads();
      {
        const uint32_t m0 = s_min[0][lane], m1 = s_min[1][lane];
        mn = m0 < m1 ? m0 : m1;
      }
      uint32_t rlt = 0, rle = 0;  // lane minima below / at or below this lane's: the order statistics j in [rlt, rle) are this lane's value
#pragma unroll
      for (int l = 0; l < kWave; ++l) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)mn, l);
        rlt += o < mn ? 1u : 0u;
        rle += o <= mn ? 1u : 0u;
      }
      auto ostat = [&](uint32_t j) {  // the (j + 1)-th smallest lane minimum
        const uint64_t m = __ballot(rlt <= j && j < rle);
        return (uint32_t)__builtin_amdgcn_readlane((int)mn, m ? __ffsll((unsigned long long)m) - 1 : 0);
      };
      auto value = [](uint32_t bits) { return __uint_as_float(order_bits_to_f32_bits(bits)); };
      uint32_t hi = ostat(PS - 1u);
      if (hi == 0xFFFFFFFFu) serial = true;  // fewer than PS lanes hold a number at all: NaNs would have to fill the list -- the ordered inserts handle it
      else {
        // invariant (order bits): count(<= hi) = c >= PS, and the PS-th smallest value is >= lo.  Split points by regula falsi
        // in the VALUE domain between an anchor below the target (first a lane minimum with its EXPECTED count -- 64 values
        // per lane, evenly dealt: count(<= the j-th smallest minimum) ~ -64 ln(1 - j / 64) -- then the last split point that
        // counted short) and hi, aiming at a count midway between PS and 64; clamped into [lo, hi - 1] in the bit domain, so
        // that every step shrinks the bracket whatever the values are.
        uint32_t c0 = 0;
        uint32_t lo = ostat(0u), c = count_le(hi, c0);
        const uint32_t ja = PS / 2u;
        float af = value(ostat(ja - 1u)), ac = -64.0f * __logf(1.0f - (float)ja * (1.0f / 64.0f));
        const float target = 0.5f * (float)(PS + (uint32_t)kWave);
        while (c > (uint32_t)kWave && lo < hi) {
          const float hf = value(hi);
          float frac = (target - ac) / ((float)c - ac);
          frac = frac < 0.1f ? 0.1f : (frac > 0.9f ? 0.9f : frac);
          uint32_t mid = f32_to_order_bits(af + (hf - af) * frac);
          if (!(mid >= lo && mid < hi)) mid = lo + ((hi - lo) >> 1);  // inf / NaN / rounding / an anchor off the bracket: plain bisection step
          uint32_t cm0 = 0;
          const uint32_t cm = count_le(mid, cm0);
          if (cm >= PS) { hi = mid; c = cm; c0 = cm0; }
          else { lo = mid + 1; af = value(mid); ac = (float)cm; }
        }
        T = hi; cT = c; c0T = c0;
        serial = c > (uint32_t)kWave;  // more than 64 values at or below the PS-th: the ordered inserts decide by index
      }
    } else {
      cT = count_if([&](int r) { return idx_of(r) < k; }, c0T);
    }
    if (stamps && n0 == 0) tb = __builtin_amdgcn_s_memtime();
    uint64_t chunk;  // the chunk's 64 smallest keys, ascending over the lanes, in both waves
    if (!serial) {
      uint32_t base = wid == 0 ? 0u : c0T;
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        const bool in = gb[r] <= T && idx_of(r) < k;
        const uint64_t m = __ballot(in);
        if (in) s_keys[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = ((uint64_t)gb[r] << 32) | idx_of(r);
        base += (uint32_t)__popcll(m);
      }
      __syncthreads();  // (the next chunk's keys are written behind at least one more barrier)
      cur = (uint32_t)lane < cT ? s_keys[lane] : kKeyMax;
      wave_rank_sort64(cur, lane);  // (unique: they carry their index)
      chunk = cur;
    } else {
      uint64_t mine = kKeyMax;
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        const uint32_t n = idx_of(r);
        wave_topk_update(mine, kWave, n < k ? (((uint64_t)gb[r] << 32) | n) : kKeyMax, kKeyMax);
      }
      s_x[wid][lane] = mine;
      __syncthreads();
      const uint64_t other = s_x[wid ^ 1][lane];
      __syncthreads();
      wave_merge_sorted64(mine, other, lane);
      chunk = mine;
    }
    if (n0 == 0) sel = chunk;
    else wave_merge_sorted64(sel, chunk, lane);  // the 64 smallest of (best so far, this chunk)
  }
  // (lanes >= PS hold larger keys of the last merge: not candidates)
  if (lane >= (int)PS) sel = kKeyMax;
  const uint32_t n_sel = PS < k ? PS : k;
  const float tau = __uint_as_float(order_bits_to_f32_bits((uint32_t)(readlane64(sel, (int)n_sel - 1) >> 32)));
  const unsigned long long t1 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
  // (2) exact re-score (both waves compute E and the candidates worth their rows; the rows are dealt to them below).
  // |q|^2 first (any summation order: E covers d u |q|^2, a tree of partial sums errs less), because E decides WHICH
  // candidates are worth their 3 KB row: with g_P the P-th smallest G, the P nearest-by-G candidates all have
  // D_ref <= g_P + |q|^2 + E, so the P-th exact distance is at most that, while a candidate with G > g_P + 2E has
  // D_ref >= G + |q|^2 - E > g_P + |q|^2 + E: strictly behind P others -- it cannot be among the top P and is not read.
  // (Round 2 re-scored all P + 16: 151 MB of centroid rows per batch from the Infinity Cache at cfg3; ~P + 3 are needed.)
  const float* qv = qp + (uint64_t)q * ldq;
  float qn = 0.0f;
  for (uint32_t j = 4u * (uint32_t)lane; j < ldc; j += 4u * kWave) {
    const f32x4 q4 = *reinterpret_cast<const f32x4*>(qv + j);
    qn += q4[0] * q4[0] + q4[1] * q4[1] + q4[2] * q4[2] + q4[3] * q4[3];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) qn += __shfl_xor(qn, off, kWave);
  const float E = ((5.0f * (float)d_pad + 16.0f) * 5.9604645e-08f + kX3Slack) * (qn + cmax2 + (metric ? 1.0f : 0.0f));  // (+ the bf16x3 product's share)
  const uint32_t Pq = P < k ? P : k;
  const float gP = __uint_as_float(order_bits_to_f32_bits((uint32_t)(readlane64(sel, (int)Pq - 1) >> 32)));
  const float gl = __uint_as_float(order_bits_to_f32_bits((uint32_t)(sel >> 32)));
  const bool have = lane < (int)n_sel && !(gl > gP + 2.0f * E);  // (negated: NaN / inf anywhere keeps the candidate)
  const uint32_t ci = have ? (uint32_t)sel : (uint32_t)readlane64(sel, 0);  // idle lanes re-read lane 0's row (one broadcast line)
  // The candidates' rows are read coalesced, four rows per load instruction, their products parked in LDS and every lane walks
  // its own candidate's chain over them (staged.hip.h).  (Rounds 2-4: lane l walked centroid row l straight from the Infinity
  // Cache, 16 bytes per load and lane -- 64 lines per instruction: 58 k of the kernel's 120 k cycles per query.)
  // The candidates are sorted by G, so the ones worth their row are a prefix of the lanes (NaN keys sort last and are kept: a gap
  // between them and the prefix is staged unused).
  // Candidate j is re-scored by wave j & 1 as its staged row j >> 1.
  const uint64_t hm = __ballot(have);
  const int n_rows = hm ? 64 - __builtin_clzll((unsigned long long)hm) : 0;
  float acc;
  {
    auto run = [&](auto nl_tag) {
      constexpr int NL = decltype(nl_tag)::value;
      const float* rp[NL];
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int j = 2 * (4 * i + (lane >> 4)) + wid;
        const uint32_t c = (uint32_t)__shfl((int)ci, j < kWave ? j : 0, kWave);
        rp[i] = C_rm + (uint64_t)c * ldc + 4 * (lane & 15);
      }
      const float* ql = qv + 4 * (lane & 15);
      constexpr int D = 2;  // chunks in flight (centroid rows: Infinity-Cache hits)
      constexpr uint32_t kBuf = (uint32_t)staged_lds_floats(kSelRowsW);
      return metric == 0 ? staged_chains<NL, 0, D, false, 4, 2>(rp, 4u, ql, ldc, s_prod, kBuf, lane)
                         : staged_chains<NL, 1, D, false, 4, 2>(rp, 4u, ql, ldc, s_prod, kBuf, lane);
    };
    if (n_rows <= 40) acc = run(std::integral_constant<int, 5>{});  // <= 20 rows per wave
    else acc = run(std::integral_constant<int, kSelRowsW / 4>{});
  }
  if (metric) acc = __fsub_rn(1.0f, acc);  // cosine distance: 1 - dot (base.rs:153-155)
  {  // lane l < 32 of wave w holds candidate 2 l + w's chain: the exact keys meet in LDS, wave 0 goes on alone
    const int j = (2 * lane + wid) & (kWave - 1);
    const bool mine_have = lane < kWave / 2 && ((hm >> j) & 1ull) != 0;
    const uint32_t cj = (uint32_t)__shfl((int)ci, j, kWave);
    if (lane < kWave / 2) s_x[0][j] = mine_have ? make_key(acc, cj) : kKeyMax;
    if (__ballot(mine_have && acc != acc) != 0 && lane == 0) atomicOr(status, 1u);
  }
  __syncthreads();
  if (wid != 0) return;
  bool nan_seen = false;
  const unsigned long long t2 = stamps ? __builtin_amdgcn_s_memtime() : 0ull;
  uint64_t exact = s_x[0][lane];                     // sorted by (exact distance, index)
  wave_rank_sort64(exact, lane);                     // (~35 ordered inserts of ~150 cycles each before)
  // (3) certificate
  const float dP = __uint_as_float(order_bits_to_f32_bits((uint32_t)(readlane64(exact, (int)Pq - 1) >> 32)));
  // metric 1: G ~ D_ref - 1 with |D_ref - (1 + G)| <= u (1 + 2 |q||c|) + 3.03 d u |q||c| < (5d + 16) u (|q|^2 + max|c|^2 + 1)
  const bool certified = (n_sel >= k) || (dP < tau + (metric ? 1.0f : qn) - E);  // NaN anywhere -> false -> exact path decides
  if (!certified) {
    // exact fallback for this query: every centroid, ordered chain per lane
    if (lane == 0) atomicAdd(fallback_count, 1u);
    exact = kKeyMax;
    for (uint32_t n0 = 0; n0 < k; n0 += kWave) {
      const uint32_t n = n0 + lane;
      float a2 = 0.0f;
      if (n < k) {
        const float* cc = C_rm + (uint64_t)n * ldc;
        for (uint32_t j = 0; j < ldc; j += 4) {
          const f32x4 c4 = *reinterpret_cast<const f32x4*>(cc + j);
          const f32x4 q4 = *reinterpret_cast<const f32x4*>(qv + j);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (metric == 0) {
              const float t = __fsub_rn(c4[u], q4[u]);
              a2 = __fadd_rn(a2, __fmul_rn(t, t));
            } else {
              a2 = __fadd_rn(a2, __fmul_rn(c4[u], q4[u]));
            }
          }
        }
        if (metric) a2 = __fsub_rn(1.0f, a2);
        nan_seen |= a2 != a2;
      }
      wave_topk_update(exact, Pq, n < k ? make_key(a2, n) : kKeyMax, kKeyMax);
    }
  }
  if (lane < (int)P) probe[(uint64_t)q * P + lane] = lane < (int)Pq ? exact : kKeyMax;
  if (__ballot(nan_seen) != 0 && lane == 0) atomicOr(status, 1u);
  if (pq.b) {  // P <= 48 here: one chunk
    uint32_t carry = 0, n_visited = 0;
    plan_query_chunk(pq, q, lane, 0u, lane < (int)Pq ? exact : kKeyMax, carry, n_visited);
    plan_query_finish(pq, q, lane, carry, n_visited);
  }
  if (stamps && lane == 0) {
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    atomicAdd(stamps + 24, t1 - t0); atomicAdd(stamps + 25, t2 - t1); atomicAdd(stamps + 26, t3 - t2); atomicAdd(stamps + 27, 1ull);
    atomicAdd(stamps + 28, ta - t0); atomicAdd(stamps + 29, tb - ta); atomicAdd(stamps + 30, t1 - tb);  // of the selection: loads + bits, bracket, compact + sort
  }
}

// ---- the same selection for MORE ranked lists than one key per lane holds (round 6: nprobe in (48, 200]) -----------------------------------
// coarse_select_rescore_kernel keeps P + 16 candidates, a key per lane; beyond 48 ranked lists the batched coarse quantiser went back to
// the ordered chains, 64 ranks per pass (~450 us per 256 queries at nprobe = 128, cfg3).  Here the candidates are a WIDE list (wide.hip.h:
// 256 keys, four per lane): a block of four waves per query, every wave folds its quarter of the row's values 256 at a time (sort, merge
// with the best so far), the four lists are folded, the PS = P + 32 smallest are re-scored exactly -- a thread per candidate walks its
// centroid's ordered chain --, ranked by counting and certified exactly like the narrow kernel's; a query that fails is re-ranked over
// ALL centroids with exact keys through the same fold.  probe[q][0..P) and, with pq.b != 0, the query's plan come out the same way.
constexpr int kSelWideWaves = 4;
static __global__ __launch_bounds__(kWave * kSelWideWaves) void coarse_select_wide_kernel(
    const float* G, uint32_t N_pad, uint32_t k, const float* C_rm, uint32_t ldc, const float* qp, uint32_t ldq, uint32_t d_pad,
    float cmax2, uint32_t P, uint32_t PS, uint64_t* probe, uint32_t* status, uint32_t* fallback_count, int metric, PlanQ pq, uint32_t n_queries) {
  __shared__ uint64_t sh[kSelWideWaves][kWideR][kWave];
  __shared__ uint64_t s_key[kWideKeys];
  __shared__ uint32_t s_ci[kWideKeys];
  __shared__ float s_qn[kSelWideWaves];
  __shared__ uint32_t s_nrows, s_fail;
  __shared__ float s_tau;
  extern __shared__ __attribute__((aligned(16))) float qs[];  // the query [ldc]
  const uint32_t q = blockIdx.x;
  if (q >= n_queries) return;  // (whole blocks)
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* g = G + (uint64_t)q * N_pad;
  const float* qv = qp + (uint64_t)q * ldq;
  float qpart = 0.0f;
  for (uint32_t j = threadIdx.x; j < ldc; j += blockDim.x) {
    const float v = qv[j];
    qs[j] = v;
    qpart += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) qpart += __shfl_xor(qpart, off, kWave);
  if (lane == 0) s_qn[wid] = qpart;
  // the 256 smallest keys key_of(n), n < k, ascending, in wave 0 (every wave: its share 256 values at a time, then the fold)
  auto smallest = [&](auto&& key_of, uint64_t (&best)[kWideR]) {
#pragma unroll
    for (int r = 0; r < kWideR; ++r) best[r] = kKeyMax;
    for (uint32_t n0 = (uint32_t)wid * kWideKeys; n0 < k; n0 += kSelWideWaves * kWideKeys) {  // (wave-uniform)
      uint64_t c[kWideR];
#pragma unroll
      for (int r = 0; r < kWideR; ++r) {
        const uint32_t n = n0 + (uint32_t)(r * kWave + lane);
        c[r] = n < k ? key_of(n) : kKeyMax;
      }
      wide_sort(c, lane);
      if (readlane64(c[0], 0) >= readlane64(best[kWideR - 1], kWave - 1)) continue;
      wide_merge_sorted(best, c, lane);
    }
    __syncthreads();  // (`sh` may still be read from a previous fold)
#pragma unroll
    for (int r = 0; r < kWideR; ++r) sh[wid][r][lane] = best[r];
    __syncthreads();
#pragma unroll
    for (int s2 = 1; s2 < kSelWideWaves; s2 <<= 1) {
      if ((wid & (2 * s2 - 1)) == 0) {
        uint64_t o[kWideR];
#pragma unroll
        for (int r = 0; r < kWideR; ++r) o[r] = sh[wid + s2][r][lane];
        wide_merge_sorted(best, o, lane);
        if (2 * s2 < kSelWideWaves && wid != 0) {
#pragma unroll
          for (int r = 0; r < kWideR; ++r) sh[wid][r][lane] = best[r];
        }
      }
      if (2 * s2 < kSelWideWaves) __syncthreads();
    }
  };
  uint64_t sel[kWideR];
  smallest([&](uint32_t n) { return ((uint64_t)f32_to_order_bits(g[n]) << 32) | n; }, sel);
  const uint32_t n_sel = PS < k ? PS : k, Pq = P < k ? P : k;
  float qn = 0.0f;
  for (int w = 0; w < kSelWideWaves; ++w) qn += s_qn[w];  // (written before the fold's barriers)
  const float E = ((5.0f * (float)d_pad + 16.0f) * 5.9604645e-08f + kX3Slack) * (qn + cmax2 + (metric ? 1.0f : 0.0f));
  if (wid == 0) {
    // which candidates are worth their row (coarse_select_rescore_kernel (2)): G <= g_P + 2 E; sorted by G, they are a prefix
    const float gP = __uint_as_float(order_bits_to_f32_bits((uint32_t)(wide_get(sel, Pq - 1u) >> 32)));
    uint32_t n_rows = 0;
#pragma unroll
    for (int r = 0; r < kWideR; ++r) {
      const uint32_t e = (uint32_t)(r * kWave + lane);
      const float gl = __uint_as_float(order_bits_to_f32_bits((uint32_t)(sel[r] >> 32)));
      const bool have = e < n_sel && sel[r] != kKeyMax && !(gl > gP + 2.0f * E);  // (negated: NaN / inf anywhere keeps the candidate)
      const uint64_t hm = __ballot(have);
      if (hm) n_rows = (uint32_t)(r * kWave) + 64u - (uint32_t)__builtin_clzll((unsigned long long)hm);
      s_ci[e] = (uint32_t)sel[r];
    }
    const float tau = __uint_as_float(order_bits_to_f32_bits((uint32_t)(wide_get(sel, n_sel - 1u) >> 32)));  // (every lane: wide_get reads a lane of a register)
    if (lane == 0) {
      s_nrows = n_rows;
      s_tau = tau;
    }
  }
  __syncthreads();
  const uint32_t n_rows = s_nrows;
  auto exact_of = [&](uint32_t n) {  // the reference's ordered chain of centroid n against the query (ivfflat.rs:159, base.rs:119-126)
    const f32x4* cc = reinterpret_cast<const f32x4*>(C_rm + (uint64_t)n * ldc);
    const f32x4* q4p = reinterpret_cast<const f32x4*>(qs);
    float a2 = 0.0f;
#pragma unroll 8
    for (uint32_t j = 0; j < ldc / 4; ++j) {
      const f32x4 c4 = cc[j], q4 = q4p[j];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (metric == 0) {
          const float t = __fsub_rn(c4[u], q4[u]);
          a2 = __fadd_rn(a2, __fmul_rn(t, t));
        } else {
          a2 = __fadd_rn(a2, __fmul_rn(c4[u], q4[u]));
        }
      }
    }
    if (metric) a2 = __fsub_rn(1.0f, a2);
    if (a2 != a2) atomicOr(status, 1u);
    return make_key(a2, n);
  };
  uint64_t exact = kKeyMax;
  if (threadIdx.x < n_rows) exact = exact_of(s_ci[threadIdx.x]);
  s_key[threadIdx.x] = exact;
  __syncthreads();
  uint32_t rank = 0xFFFFFFFFu;
  if (threadIdx.x < n_rows && exact != kKeyMax) {
    rank = 0;
    for (uint32_t j = 0; j < n_rows; ++j) rank += s_key[j] < exact ? 1u : 0u;
  }
  __syncthreads();  // every thread has read the unranked keys
  s_key[threadIdx.x] = kKeyMax;
  __syncthreads();
  if (rank < kWideKeys) s_key[rank] = exact;  // (unique keys: unique ranks)
  __syncthreads();
  // (3) certificate: every unselected centroid has D_ref >= tau + |q|^2 - E
  if (threadIdx.x == 0) {
    const float dP = __uint_as_float(order_bits_to_f32_bits((uint32_t)(s_key[Pq - 1u] >> 32)));
    const bool certified = (n_sel >= k) || (s_key[Pq - 1u] != kKeyMax && dP < s_tau + (metric ? 1.0f : qn) - E);  // NaN anywhere -> false
    s_fail = certified ? 0u : 1u;
    if (!certified) atomicAdd(fallback_count, 1u);
  }
  __syncthreads();
  if (s_fail) {  // (block-uniform) every centroid's exact key through the same fold
    uint64_t ex[kWideR];
    smallest(exact_of, ex);
    if (wid == 0) {
#pragma unroll
      for (int r = 0; r < kWideR; ++r) s_key[r * kWave + lane] = ex[r];
    }
    __syncthreads();
  }
  for (uint32_t j = threadIdx.x; j < P; j += blockDim.x) probe[(uint64_t)q * P + j] = j < Pq ? s_key[j] : kKeyMax;
  if (pq.b && wid == 0) {  // the query's plan (plan.hip.h step 1): lane j of chunk c = probe rank 64 c + j
    uint32_t carry = 0, n_visited = 0;
    for (uint32_t c0 = 0; c0 < P; c0 += kWave) {
      const uint32_t j = c0 + (uint32_t)lane;
      plan_query_chunk(pq, q, lane, c0, j < Pq ? s_key[j] : kKeyMax, carry, n_visited);
    }
    plan_query_finish(pq, q, lane, carry, n_visited);
  }
}

// ---- k-means assign through the matrix cores (ivfflat.rs:29-46) -----------------------------------------
// dist_gemm_kernel<true> leaves, per (tile of 128 centroids, point), the best / second-best approximate value
// |c|^2 - 2 <x, c> (or -<x, c>) and the best's centroid.  Thread per point: fold the k/128 triples in ascending
// centroid order (coalesced: consecutive threads read consecutive points of one tile row).
static __global__ void assign_argmin_merge_kernel(const float* part_v1, const uint32_t* part_c1, const float* part_v2, uint32_t n_tiles,
                                                  uint32_t n_pad, uint32_t nb, uint32_t* best, float* g2, uint32_t* q_start = nullptr,
                                                  const uint32_t* q_count = nullptr) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  // (where this batch's entries of the tile re-scan's queue will begin: assign_rescore_kernel, next on the stream, appends them)
  if (i == 0 && q_start != nullptr) *q_start = *q_count;
  if (i >= nb) return;
  float a1 = part_v1[i], a2 = part_v2[i];
  uint32_t ac = part_c1[i];
  bool any_nan = a2 != a2;
  for (uint32_t t = 1; t < n_tiles; ++t) {  // ascending centroid ranges: a later equal value never replaces the candidate
    const float b1 = part_v1[(uint64_t)t * n_pad + i], b2 = part_v2[(uint64_t)t * n_pad + i];
    any_nan |= b2 != b2;
    if (b1 < a1) {
      a2 = a1 < b2 ? a1 : b2;
      a1 = b1;
      ac = part_c1[(uint64_t)t * n_pad + i];
    } else {
      const float m = b1 < a2 ? b1 : a2;  // includes b1 == a1: second == best, not certified
      a2 = m;
    }
  }
  best[i] = ac;
  g2[i] = any_nan ? __builtin_nanf("") : a2;  // NaN / inf never certify (assign_rescore_kernel)
}

// Exact D(x_i, c_best) in reference arithmetic (one lane per point) + certificate: every other centroid has
// approximate value >= g2, hence exact distance >= g2 + |x|^2 - E; if the candidate's exact distance is strictly
// below that it is the unique first minimum.  Otherwise the point is queued for the exact scan.
// (Round 6, second half: the 64 points of a block -- ONE wave -- fetch their rows and their candidates' rows TOGETHER, 32 columns at a
// time in whole 128-byte lines, eight rows per load instruction, through a wave-private LDS image from which every lane then walks its own
// row's chain in the reference's order.  Before, a lane read its own rows 16 bytes at a time: 64 lines per load instruction, each line
// fetched eight times over because 8 waves x 128 live lines do not fit the L1 -- 242 us per 131072 points, a quarter of the assign pass.)
constexpr int kRsChunk = 32;  // columns per step
constexpr int kRsPitch = 36;  // floats per staged row: lane-per-row ds_read_b128 without bank conflicts
static __global__ __launch_bounds__(kWave) void assign_rescore_kernel(const float* X, uint32_t ldx, const float* C_rm, uint32_t ldc, uint32_t d, uint32_t d_pad,
                                             const float* cmax2_dev, const uint32_t* best, const float* g2, uint32_t nb, uint32_t k,
                                             uint32_t i_base, uint32_t* assign, float* mind, uint32_t* fb_list, uint32_t* fb_count,
                                             uint32_t* status, int metric, float* fb_thr = nullptr, const uint32_t* rc2_bits = nullptr) {
  // rc2_bits != nullptr: the values came from the SINGLE fp16 product (the cascade's first filter): the bound uses the measured residuals
  __shared__ __attribute__((aligned(16))) float sx[kWave * kRsPitch];
  __shared__ __attribute__((aligned(16))) float sc[kWave * kRsPitch];
  const int lane = threadIdx.x;  // (a block is one wave: no barrier anywhere, the LDS image is the wave's own and its accesses are in order)
  const uint32_t i_raw = blockIdx.x * kWave + (uint32_t)lane;
  const bool live = i_raw < nb;
  const uint32_t i = live ? i_raw : nb - 1;  // (dead lanes of the last block walk the last point's rows along and drop the result)
  const uint32_t c = best[i];
  const float* x = X + (uint64_t)i * ldx;
  const float* cv = C_rm + (uint64_t)c * ldc;
  const bool want_r = rc2_bits != nullptr;
  float acc = 0.0f, xn = 0.0f, rx2 = 0.0f;
  auto element = [&](float xv, float cvv) {
    if (metric == 0) {
      const float t = __fsub_rn(xv, cvv);  // data_point.squared_euclidean(centroid): ivfflat.rs:37
      acc = __fadd_rn(acc, __fmul_rn(t, t));
    } else {
      acc = __fadd_rn(acc, __fmul_rn(xv, cvv));
    }
    xn = __fadd_rn(xn, __fmul_rn(xv, xv));
    if (want_r) { const float dl = xv - (float)(_Float16)xv; rx2 = __fadd_rn(rx2, __fmul_rn(dl, dl)); }
  };
  uint32_t j = 0;
  const bool vec_ok = ((ldx | ldc) & 3u) == 0 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(C_rm)) & 15u) == 0;
  if (vec_ok && d >= (uint32_t)kRsChunk) {  // (block-uniform)
    const uint32_t d_st = d & ~(uint32_t)(kRsChunk - 1);
    const int lrow = lane >> 3, lcol = (lane & 7) * 4;  // piece q of a step: rows 8 q + lrow, this lane's 16 bytes of their 128
    uint64_t xoff[8], coff[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int src = 8 * q + lrow;
      xoff[q] = (uint64_t)(uint32_t)__shfl((int)i, src, kWave) * ldx + (uint32_t)lcol;
      coff[q] = (uint64_t)(uint32_t)__shfl((int)c, src, kWave) * ldc + (uint32_t)lcol;
    }
    f32x4 rxv[8], rcv[8];
    auto gload = [&](uint32_t j0) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        rxv[q] = *reinterpret_cast<const f32x4*>(X + xoff[q] + j0);
        rcv[q] = *reinterpret_cast<const f32x4*>(C_rm + coff[q] + j0);
      }
    };
    gload(0);
    for (; j < d_st; j += kRsChunk) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        *reinterpret_cast<f32x4*>(sx + (8 * q + lrow) * kRsPitch + lcol) = rxv[q];
        *reinterpret_cast<f32x4*>(sc + (8 * q + lrow) * kRsPitch + lcol) = rcv[q];
      }
      if (j + kRsChunk < d_st) gload(j + kRsChunk);  // the next step's lines: in flight under this step's chain
#pragma unroll
      for (int cc = 0; cc < kRsChunk / 4; ++cc) {
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(sx + lane * kRsPitch + 4 * cc);
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(sc + lane * kRsPitch + 4 * cc);
#pragma unroll
        for (int u = 0; u < 4; ++u) element(x4[u], c4[u]);
      }
    }
  }
  if (vec_ok) {
    for (; j + 4 <= d; j += 4) {
      const f32x4 x4 = *reinterpret_cast<const f32x4*>(x + j);
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(cv + j);
#pragma unroll
      for (int u = 0; u < 4; ++u) element(x4[u], c4[u]);
    }
  }
  for (; j < d; ++j) element(x[j], cv[j]);
  if (!live) return;
  if (metric) acc = __fsub_rn(1.0f, acc);
  const float tau = g2[i];
  float E = ((5.0f * (float)d_pad + 16.0f) * 5.9604645e-08f + (rc2_bits ? 0.0f : kX3Slack)) * (xn + *cmax2_dev + (metric ? 1.0f : 0.0f));
  if (rc2_bits) {  // 2 (r_x |c| + (|x| + r_x) R_c), every factor rounded up by 1 % (the sums of squares, the roots)
    const float rx = 1.01f * __builtin_sqrtf(rx2), Rc = 1.01f * __builtin_sqrtf(__uint_as_float(*rc2_bits));
    E += 2.02f * (rx * __builtin_sqrtf(*cmax2_dev) + (__builtin_sqrtf(xn) + rx) * Rc);
  }
  const float lower = tau + (metric ? 1.0f : xn) - E;  // NaN if anything overflowed
  const bool finite = tau < __builtin_inff() && E < __builtin_inff();
  const bool certified = k == 1 || (finite && acc < lower);
  assign[i] = c;
  if (mind) mind[i] = acc;
  if (!certified) {
    const uint32_t pos = atomicAdd(fb_count, 1u);
    fb_list[pos] = i + i_base;
    // (for assign_tile_rescan_kernel) the true first minimum c* has D(c*) <= acc, hence G(c*) <= acc - |x|^2 + E: only centroid
    // tiles whose smallest G is at or below that can hold it.  NaN / inf: no tile is excluded.
    if (fb_thr) fb_thr[pos] = (acc - (metric ? 1.0f : xn)) + 1.01f * E;
  } else if (acc != acc) atomicOr(status, 1u);
}

// Uncertified points at LARGE k (cfg5: 65536 centroids): the exact re-scan of a point against all centroids costs k chains,
// but the per-tile minima of G that the contraction's epilogue left (part_v1, still in place for this batch) say which tiles
// of 128 centroids can hold the point's first minimum at all -- one or two of 512.  One wave per queued point of this batch:
// the candidate tiles' centroids get their exact ordered chains (lane per centroid), first minimum by (distance, index).
// More than kRescanTiles candidate tiles (forced-failure tests, degenerate data) or a NaN: the point goes to the full exact
// scan like before (fb2).  At k = 65536 the full scan of the ~3 % uncertified points was 19 % of a k-means pass.
constexpr uint32_t kRescanTiles = 8;
// (Round 6: the candidate tiles are read from the centroids in the SCAN layout -- lane-transposed 64-row tiles, 1 KiB wave loads, the
// point as the scalar operand: scan_item, the engine of every exact scan -- instead of a lane per centroid walking its own row-major row
// 16 bytes at a time, 64 lines per load instruction: ~100 us per candidate tile, which made this kernel as expensive as the contraction
// once the fp16 single-product filter of the assign cascade sent it a few per cent of every batch.)
struct RescanSrc {
  static constexpr bool kSeqIds = false;
  static constexpr bool kStreamOnce = false;
  uint64_t* out_ptr;
  uint32_t seq0;
  __device__ __forceinline__ uint32_t seq_base(uint32_t, int) const { return seq0; }
  __device__ __forceinline__ const uint32_t* seq_ids(uint32_t) const { return nullptr; }
  __device__ __forceinline__ uint64_t* out(uint32_t, int) const { return out_ptr; }
  __device__ __forceinline__ uint32_t bound_slot(uint32_t, int) const { return 0; }
};
// Cb: the centroids in the scan layout [tiles of 64][ld]; Xb: THIS batch's rows, zero padded to ld columns (pitch ld), row i = point i_base + i
// A BLOCK of four waves per queued point: every wave lists the point's candidate tiles (the same list in each), wave w walks the
// 64-centroid halves w, w + 4, ... of them, the block's first minimum is the smallest of the waves' keys.  (One wave per point walked
// up to sixteen halves one after the other, 7 us each -- the dependent chain of 768 adds and a 192 KB tile -- with two waves per SIMD
// on the chip.  Same box, assign pass N = 4M k = 4096: 48.0 ms with one wave per point, 46.8 with two, 46.3 with four.)
constexpr int kRescanWaves = 4;
static __global__ __launch_bounds__(kWave * kRescanWaves) void assign_tile_rescan_kernel(
    const float* Xb, const float* Cb, uint32_t ld, uint32_t k, const float* part_v1, uint32_t n_tiles,
    uint32_t pitch, uint32_t i_base, uint32_t nb, const uint32_t* fb_list, const float* fb_thr, const uint32_t* fb_count, uint32_t* assign,
    float* mind, uint32_t* fb2_list, uint32_t* fb2_count, int metric, const uint32_t* fb_start, const uint32_t* part_c1, const float* part_v2) {
  __shared__ uint64_t s_out[kRescanWaves][kWave];
  __shared__ uint64_t s_best[kRescanWaves];
  __shared__ uint32_t s_nan[kRescanWaves];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t n_q = *fb_count;
  // this batch's entries: the tail of the queue from *fb_start on (assign_argmin_merge_kernel)
  const uint32_t e0 = *fb_start;
  ScanParams p;
  p.ld = ld; p.n_chunks = ld / kChunk; p.k = 1; p.status = nullptr; p.bounds = nullptr; p.lower = nullptr; p.debug = 0; p.next_quad = nullptr; p.stamps = nullptr;
  for (uint32_t e = e0 + blockIdx.x; e < n_q; e += gridDim.x) {  // (block-uniform)
    const uint32_t idx = fb_list[e];
    if (idx < i_base || idx >= i_base + nb) continue;  // (another batch's entry: its tile minima are gone)
    const uint32_t i = idx - i_base;
    const float T = fb_thr[e];
    // the candidate tiles, candidate j in lane j (every wave makes the same list)
    uint32_t n_cand = 0, cand = 0;
    for (uint32_t t0 = 0; t0 < n_tiles && n_cand <= kRescanTiles; t0 += kWave) {
      const uint32_t t = t0 + (uint32_t)lane;
      const float v = t < n_tiles ? part_v1[(uint64_t)t * pitch + i] : __builtin_inff();
      uint64_t m = __ballot(t < n_tiles && !(v > T));  // (NaN on either side: candidate)
      while (m && n_cand <= kRescanTiles) {
        const uint32_t tl = (uint32_t)__ffsll((unsigned long long)m) - 1u;
        m &= m - 1;
        if (lane == (int)n_cand) cand = t0 + tl;
        ++n_cand;
      }
    }
    uint64_t best = kKeyMax;
    bool nan = false;
    if (n_cand <= kRescanTiles) {
      // Which HALF of a candidate tile: the one that holds the tile's smallest G (part_c1 says where) always; the other one only if the
      // tile's SECOND smallest G is not above T either -- every G of the tile but the smallest is at or above the second smallest, so a
      // half without the smallest and with that bound above T cannot hold the point's first minimum.  (Most open points have their two
      // close centroids in two tiles, or in one half.  Same box, assign pass N = 4M k = 4096: 45.6 -> 44.1 ms.)
      uint32_t half_best = 0, other_too = 1;
      if ((uint32_t)lane < n_cand && part_v2 != nullptr) {
        const float v2 = part_v2[(uint64_t)cand * pitch + i];
        const uint32_t c1 = part_c1[(uint64_t)cand * pitch + i];
        half_best = (c1 - cand * (uint32_t)kGemmBM) >= (uint32_t)kWave ? 1u : 0u;  // (c1 lies in the tile: the epilogue's arg-min)
        other_too = !(v2 > T) ? 1u : 0u;  // (NaN: both halves)
      }
      for (uint32_t u = (uint32_t)wid; u < 2u * n_cand; u += kRescanWaves) {  // (wave-uniform)
        const int src_lane = (int)(u >> 1);
        const uint32_t hb = (uint32_t)__shfl((int)half_best, src_lane, kWave), ot = (uint32_t)__shfl((int)other_too, src_lane, kWave);
        if ((u & 1u) != hb && ot == 0u) continue;
        const uint32_t c0 = (uint32_t)__shfl((int)cand, src_lane, kWave) * kGemmBM + (u & 1u) * kWave;  // a whole scan tile of 64 centroids
        if (c0 >= k) continue;
        ItemView<1> iv;
        iv.rows = Cb + (uint64_t)c0 * ld;
        iv.nrows = k - c0 < (uint32_t)kWave ? k - c0 : (uint32_t)kWave;
        iv.nq = 1;
        iv.qb = Xb + (uint64_t)i * ld;
        RescanSrc src{s_out[wid], c0};
        if (metric == 0) scan_item<1, 1, 0>(src, p, 0u, iv, lane, nan);
        else scan_item<1, 1, 1>(src, p, 0u, iv, lane, nan);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint64_t key = s_out[wid][0];  // the half tile's first minimum by (distance, centroid index)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        best = key < best ? key : best;
      }
    }
    const bool wave_nan = __ballot(nan) != 0;  // (the whole wave votes)
    if (lane == 0) { s_best[wid] = best; s_nan[wid] = wave_nan ? 1u : 0u; }
    __syncthreads();
    if (threadIdx.x == 0) {
      bool any_nan = false;
#pragma unroll
      for (int w = 0; w < kRescanWaves; ++w) { best = s_best[w] < best ? s_best[w] : best; any_nan |= s_nan[w] != 0; }
      const bool defer = n_cand > kRescanTiles || n_cand == 0 || any_nan;
      if (defer || best == kKeyMax) fb2_list[atomicAdd(fb2_count, 1u)] = idx;
      else {
        assign[idx] = (uint32_t)best;
        if (mind) mind[idx] = __uint_as_float(order_bits_to_f32_bits((uint32_t)(best >> 32)));
      }
    }
    __syncthreads();  // (s_best / s_nan are the next point's)
  }
}

// max over the finite-or-inf |c|^2 (NaN never raises it: such centroids fail every certificate through G)
static __global__ __launch_bounds__(256) void max_norm_kernel(const float* cnorm, uint32_t k, float* out) {
  __shared__ float sh[256];
  float m = 0.0f;
  for (uint32_t c = threadIdx.x; c < k; c += 256) m = fmaxf(m, cnorm[c]);
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}

static __global__ void gather_points_kernel(const float* X, uint32_t ldx, const uint32_t* list, uint32_t n_list, float* out) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (uint64_t)n_list * ldx) return;
  out[t] = X[(uint64_t)list[t / ldx] * ldx + t % ldx];
}
static __global__ void scatter_assign_kernel(const uint32_t* list, uint32_t n_list, const uint32_t* a_in, const float* m_in,
                                             uint32_t* assign, float* mind) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_list) return;
  assign[list[t]] = a_in[t];
  if (mind) mind[list[t]] = m_in[t];
}

}  // namespace vers
