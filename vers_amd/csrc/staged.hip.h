// staged.hip.h -- ordered f32 chains of a few GATHERED rows against one query (base.rs:119-126 / 153-155), the engine of the
// two exact re-scores behind a matrix-core pre-selection: coarse_select_rescore_kernel (gemm.hip.h: the candidates' centroid
// rows) and ivf_rescore_kernel (finish.hip.h: the surviving corpus rows).
//
// A lane that walks its own row reads 16 bytes per load from a row of its own: 64 lines per instruction, one per lane -- the
// texture path serialises them and the chain behind (sub, mul, add per column on one wave) is starved: 58 k of the selection's
// 120 k cycles per query, 74-82 % of the finish's wave cycles waiting (profiles/r03, r04).  Here the wave reads the rows
// COALESCED instead -- a quarter wave per row and 64-column chunk: 16 lanes x 16 bytes = 256 contiguous bytes, four rows per
// load instruction -- every lane forms the products (x - q)^2 (or x * q) of the piece it loaded, which depend on no order, and
// parks them in LDS; then lane l walks the strictly ordered chain acc = acc + m_j over row l's products: the same operations on
// the same operands in the same order as scan_item's chain, ONE dependent add per column.  The next chunk's pieces are in
// flight under this chunk's chain.  (coarse1_kernel, ivf_plan.hip, does the same for the whole-tile single-query case.)
#pragma once
#include "scan.hip.h"

namespace vers {

constexpr int kStageCols = 64;                 // columns per chunk (every leading dimension is a multiple of kColAlign = 64)
constexpr int kStagePitch = kStageCols + 4;    // floats per staged row: 17 sixteen-byte slots -- lane l's ds_read_b128 falls on slot
                                               // (17 l + w) mod 16 = (l + w) mod 16: each of the instruction's 16-lane groups covers all 16
constexpr size_t staged_lds_floats(int rows) { return (size_t)rows * kStagePitch; }

// rp[i]: row (4 i + lane / 16) of the wave's staged rows, already advanced to this lane's piece (lane % 16) of chunk 0; a row's
// consecutive 16-byte pieces are `xstep` floats apart (4: row-major; 256: the lane-transposed tiles).  qv: the query, at this
// lane's piece of chunk 0 (any address space the caller's loader reads: global or LDS).  s_prod: this wave's [4 NL][kStagePitch]
// floats.  ld: columns, a multiple of 64.  Returns the chain of staged row `lane` (lanes >= 4 NL: row 0's).
// One wave, no block-wide synchronisation: LDS operations of a wave execute in order; the fences keep the compiler from moving
// a lane's reads across ANOTHER lane's writes.
template <int NL, int METRIC, class QPtr>
__device__ __forceinline__ float staged_chains(const float* const (&rp)[NL], uint32_t xstep, QPtr qv, uint32_t ld, float* s_prod, int lane) {
  const int sub = lane >> 4, l16 = lane & 15;
  float* const wr = s_prod + sub * kStagePitch + 4 * l16;
  const f32x4* const rd = reinterpret_cast<const f32x4*>(s_prod + (lane < 4 * NL ? lane : 0) * kStagePitch);
  const uint32_t n_chunks = ld / kStageCols, last = n_chunks - 1;
  auto issue = [&](f32x4 (&x)[NL], f32x4& q4, uint32_t ch) {  // UNCONDITIONAL loads (a clamped chunk past the end: never used)
#pragma unroll
    for (int i = 0; i < NL; ++i) x[i] = *reinterpret_cast<const f32x4*>(rp[i] + (uint64_t)ch * 16u * xstep);
    q4 = *reinterpret_cast<const f32x4*>(qv + ch * kStageCols);
  };
  float acc = 0.0f;
  auto stage_and_chain = [&](const f32x4 (&x)[NL], const f32x4& q4) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      f32x4 m;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (METRIC == 0) {
          const float t = __fsub_rn(x[i][u], q4[u]);
          m[u] = __fmul_rn(t, t);
        } else {
          m[u] = __fmul_rn(x[i][u], q4[u]);
        }
      }
      *reinterpret_cast<f32x4*>(wr + 4 * i * kStagePitch) = m;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x4 m[kStageCols / 4];
#pragma unroll
    for (int w = 0; w < kStageCols / 4; ++w) m[w] = rd[w];
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // (the reads are issued before the next chunk's writes: in-order LDS does the rest)
#pragma unroll
    for (int w = 0; w < kStageCols / 4; ++w)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __fadd_rn(acc, m[w][u]);
  };
  f32x4 xa[NL], xb[NL], qa, qb;
  issue(xa, qa, 0);
  for (uint32_t ch = 0; ch < n_chunks; ch += 2) {
    issue(xb, qb, ch + 1 <= last ? ch + 1 : last);
    stage_and_chain(xa, qa);
    issue(xa, qa, ch + 2 <= last ? ch + 2 : last);
    if (ch + 1 < n_chunks) stage_and_chain(xb, qb);
  }
  return acc;
}

}  // namespace vers
