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
// the same operands in the same order as scan_item's chain, ONE dependent add per column.  The next chunks' pieces are in
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
// lane's piece of chunk 0.  s_prod: this wave's TWO product buffers of [4 NL][kStagePitch] floats (`buf_floats` apart).
// ld: columns, a multiple of 64.
// D (even): chunks in flight -- the pieces of chunk c + D are requested when chunk c's have been consumed.
// Q_LDS: the query sits in LDS and is read a step ahead (no registers held for the chunks in flight); otherwise (global) its
// pieces travel with the rows'.
// G, NB: a chunk's 16 sixteen-byte product pieces come back from LDS G at a time, NB groups in flight (4 G adds of ~9 cycles
// cover an LDS round trip): 4 G NB registers.
// The loop is software-pipelined over the two buffers: while lane l adds chunk c's products -- a dependent v_add_f32 issues
// every ~9 cycles, the wave's other VALU slots are free -- chunk c + 1's products are formed and parked in the other buffer
// BETWEEN the adds.  Within a step reads and writes touch different buffers; one fence pair per step orders the steps.  Whole
// groups of D chunks run branch-free; a remainder (ld / 64 not a multiple of D) takes the plain path.
// Returns the chain of staged row `lane` (lanes >= 4 NL: row 0's).
// One wave, no block-wide synchronisation: LDS operations of a wave execute in order; the fences keep the compiler from moving
// a lane's reads across ANOTHER lane's writes.
template <int NL, int METRIC, int D, bool Q_LDS, int G, int NB>
__device__ __forceinline__ float staged_chains(const float* const (&rp)[NL], uint32_t xstep, const float* qv, uint32_t ld, float* s_prod,
                                               uint32_t buf_floats, int lane) {
  static_assert(D % 2 == 0, "chunk parity = buffer parity");
  constexpr int kW = kStageCols / 4, kGroups = kW / G;
  static_assert(kW % G == 0 && NB <= kGroups, "groups of pieces");
  constexpr int QD = Q_LDS ? 1 : D;
  const int sub = lane >> 4, l16 = lane & 15;
  float* const wr = s_prod + sub * kStagePitch + 4 * l16;
  const float* const rd = s_prod + (lane < 4 * NL ? lane : 0) * kStagePitch;
  const uint32_t n_chunks = ld / kStageCols, last = n_chunks - 1;
  auto issue_x = [&](f32x4 (&x)[NL], uint32_t ch) {  // UNCONDITIONAL loads (a clamped chunk past the end: never used)
#pragma unroll
    for (int i = 0; i < NL; ++i) x[i] = *reinterpret_cast<const f32x4*>(rp[i] + (uint64_t)ch * 16u * xstep);
  };
  auto load_q = [&](uint32_t ch) { return *reinterpret_cast<const f32x4*>(qv + (ch <= last ? ch : last) * kStageCols); };
  auto product = [&](const f32x4& x, const f32x4& q4) {
    f32x4 m;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (METRIC == 0) {
        const float t = __fsub_rn(x[u], q4[u]);
        m[u] = __fmul_rn(t, t);
      } else {
        m[u] = __fmul_rn(x[u], q4[u]);
      }
    }
    return m;
  };
  auto stage = [&](const f32x4 (&x)[NL], const f32x4& q4, int buf) {
#pragma unroll
    for (int i = 0; i < NL; ++i) *reinterpret_cast<f32x4*>(wr + buf * buf_floats + 4 * i * kStagePitch) = product(x[i], q4);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  float acc = 0.0f;
  f32x4 x[D][NL], q4[QD];
#pragma unroll
  for (int s = 0; s < D; ++s) issue_x(x[s], (uint32_t)s <= last ? (uint32_t)s : last);
#pragma unroll
  for (int s = 0; s < QD; ++s) q4[s] = load_q((uint32_t)s);
  const uint32_t n_main = n_chunks - n_chunks % D;
  if (n_main != 0) {
    stage(x[0], q4[0], 0);
    for (uint32_t ch = 0; ch < n_main; ch += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {  // chunk ch + s: its products sit in buffer s % 2; chunk ch + s + 1 (past the last group: a
        const f32x4* const src = reinterpret_cast<const f32x4*>(rd + (s & 1) * buf_floats);  // clamped chunk nobody reads) goes to the other
        float* const dst = wr + ((s + 1) & 1) * buf_floats;
        const f32x4 (&xn)[NL] = x[(s + 1) % D];
        f32x4 mb[NB][G];
        auto rdg = [&](int g) {
#pragma unroll
          for (int w = 0; w < G; ++w) mb[g % NB][w] = src[G * g + w];
        };
#pragma unroll
        for (int g = 0; g < NB; ++g) rdg(g);
        f32x4 qn;
        if (Q_LDS) qn = load_q(ch + s + 1);
        else {
          qn = q4[(s + 1) % QD];
          q4[s % QD] = load_q(ch + s + D);  // (this chunk's query piece was consumed a step ago)
        }
        {
          const uint32_t nx = ch + s + D;  // this chunk's registers (consumed a step ago) take chunk ch + s + D
          issue_x(x[s], nx <= last ? nx : last);
        }
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
#pragma unroll
          for (int w = 0; w < G; ++w) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __fadd_rn(acc, mb[g % NB][w][u]);
#pragma unroll
            for (int i = (G * g + w) * NL / kW; i < (G * g + w + 1) * NL / kW; ++i)
              *reinterpret_cast<f32x4*>(dst + 4 * i * kStagePitch) = product(xn[i], qn);
          }
          if (g + NB < kGroups) rdg(g + NB);
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  for (uint32_t ch = n_main; ch < n_chunks; ++ch) {  // remainder: plain
    f32x4 xr[NL];
    issue_x(xr, ch);
    stage(xr, load_q(ch), 0);
    const f32x4* const src = reinterpret_cast<const f32x4*>(rd);
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      f32x4 m[G];
#pragma unroll
      for (int w = 0; w < G; ++w) m[w] = src[G * g + w];
#pragma unroll
      for (int w = 0; w < G; ++w)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __fadd_rn(acc, m[w][u]);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // (the reads are issued before any later write: in-order LDS does the rest)
  }
  return acc;
}

}  // namespace vers
