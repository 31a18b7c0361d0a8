// finish_wide.hip.h -- the exact finish of the matrix-core list scan for results WIDER than one key per lane (round 6): the same
// three steps as ivf_rescore_kernel (finish.hip.h) -- merge the query's partial candidate lists, certify, recompute the survivors in
// the reference's own arithmetic (ordered f32 chain, base.rs:119-126), order by the exact (distance, position) key and emit
// (ivfflat.rs:176-195) -- on lists of up to 256 keys, four per lane (wide.hip.h).  kp = top_k + slack keys per list, <= kWideMaxKp.
// Queries that fail the certificate are queued for fallback_kernel exactly like the narrow finish's.  Included by ivf_search.hip only.
#pragma once
#include "finish.hip.h"
#include "wide.hip.h"

namespace vers {

constexpr int kWideWaves = 4;
inline size_t rescore_wide_lds_bytes(uint32_t ld) { return (size_t)ld * sizeof(float); }  // the query (everything else is static)

__global__ __launch_bounds__(kWave * kWideWaves) void ivf_rescore_wide_kernel(RescoreArgs a) {
  __shared__ uint64_t sh[kWideWaves][kWideR][kWave];  // the waves' lists meet here (8 KB)
  __shared__ uint64_t s_key[kWideKeys];                // wave 0's merged approximate keys, then the survivors' exact keys
  __shared__ uint32_t s_row[kWideKeys], s_rid[kWideKeys];
  __shared__ float sred[kWideWaves], sres[kWideWaves];
  __shared__ uint32_t s_failed, s_nsurv;
  extern __shared__ __attribute__((aligned(16))) float qs[];  // the query, padded
  const uint32_t q = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float* qrow = a.qp + (uint64_t)q * a.ldq;
  const uint32_t* pl = a.pj_list + (uint64_t)q * a.P;
  const uint32_t* pp = a.pj_pref + (uint64_t)q * a.P;
  const uint32_t* nqp = a.pj_nq + (uint64_t)q * a.P;
  const uint32_t flag0 = a.qflags[(uint64_t)q * a.P];
  const uint32_t xmax_bits = *a.xmax2_bits;
  const uint64_t* keys = a.partials + (uint64_t)q * a.P * a.S_max * a.kp;
  // the query into LDS; |q|^2 and, for hi-only query blocks, the staged query's fp16 residual (any order: the bound inflates them)
  float qpart = 0.0f, rpart = 0.0f;
  const float qscale = a.metric ? -1.0f : -2.0f;
  for (uint32_t i = threadIdx.x; i < a.ld; i += blockDim.x) {
    const float v = qrow[i];
    qs[i] = v;
    qpart = __fadd_rn(qpart, __fmul_rn(v, v));
    if (a.shadow == 2) {
      const float y = qscale * v, dl = y - (float)(_Float16)y;
      rpart = __fadd_rn(rpart, __fmul_rn(dl, dl));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { qpart += __shfl_xor(qpart, off, kWave); rpart += __shfl_xor(rpart, off, kWave); }
  if (lane == 0) { sred[wid] = qpart; sres[wid] = rpart; }
  // ---- merge: wave w folds the WRITTEN slots w, w + 4, ... of every chunk of 64 probes (slot (probe j, quad s) exists iff s < pj_nq[j]);
  // a slot is kp ascending keys, four registers' worth.  A slot whose head is not below the list's last key contributes nothing.
  uint64_t list[kWideR];
#pragma unroll
  for (int r = 0; r < kWideR; ++r) list[r] = kKeyMax;
  for (uint32_t c0 = 0; c0 < a.P; c0 += kWave) {  // (block-uniform)
    const uint32_t jp = c0 + (uint32_t)lane;
    const uint32_t my_nq = jp < a.P ? (nqp[jp] < a.S_max ? nqp[jp] : a.S_max) : 0u;
    const uint32_t incl = wave_incl_u32(my_nq);
    const uint32_t excl = incl - my_nq;
    const uint32_t n_live = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
    for (uint32_t c = (uint32_t)wid; c < n_live; c += kWideWaves) {  // compact index -> (probe j, quad c - first slot of j)
      const uint64_t m = __ballot(my_nq != 0 && excl <= c);
      const int j = m ? 63 - __builtin_clzll((unsigned long long)m) : 0;
      const uint32_t e_j = (uint32_t)__builtin_amdgcn_readlane((int)excl, j);
      const uint64_t* sl = keys + (uint64_t)((c0 + (uint32_t)j) * a.S_max + (c - e_j)) * a.kp;
      uint64_t cand[kWideR];
#pragma unroll
      for (int r = 0; r < kWideR; ++r) cand[r] = (uint32_t)(r * kWave + lane) < a.kp ? sl[r * kWave + lane] : kKeyMax;
      if (readlane64(cand[0], 0) >= readlane64(list[kWideR - 1], kWave - 1)) continue;  // (wave-uniform; an empty slot: kKeyMax)
      wide_merge_sorted(list, cand, lane);
    }
  }
#pragma unroll
  for (int r = 0; r < kWideR; ++r) sh[wid][r][lane] = list[r];
  __syncthreads();
#pragma unroll
  for (int s = 1; s < kWideWaves; s <<= 1) {
    if ((wid & (2 * s - 1)) == 0) {
      uint64_t o[kWideR];
#pragma unroll
      for (int r = 0; r < kWideR; ++r) o[r] = sh[wid + s][r][lane];
      wide_merge_sorted(list, o, lane);
      if (2 * s < kWideWaves && wid != 0) {
#pragma unroll
        for (int r = 0; r < kWideR; ++r) sh[wid][r][lane] = list[r];
      }
    }
    if (2 * s < kWideWaves) __syncthreads();
  }
  // ---- certificate (wave 0), as in ivf_rescore_kernel: tau = the k-th smallest val, e_k = the largest bound among the k smallest;
  // a row can be among the true top-k only if val - e <= tau + e_k; a full list may have cut rows off: its last val must clear
  // tau + e_k + the bound of a row outside the list.
  if (wid == 0) {
    float qn = 0.0f, rq = 0.0f;
    for (int w = 0; w < kWideWaves; ++w) { qn += sred[w]; rq += sres[w]; }
    const PreBound pb = pre_bound((double)qn, (double)__uint_as_float(xmax_bits), a.shadow ? (double)__uint_as_float(a.xmax2_bits[2]) : 0.0, a.d_pad, a.metric, a.shadow, (double)rq);
    uint32_t cnt = 0;
    bool valid[kWideR];
    double e_mine[kWideR];
    float val[kWideR];
#pragma unroll
    for (int r = 0; r < kWideR; ++r) {
      if ((uint32_t)(r * kWave + lane) >= a.kp) list[r] = kKeyMax;  // (the networks keep 256 keys: the kp smallest are the candidates)
      valid[r] = list[r] != kKeyMax;
      cnt += (uint32_t)__popcll(__ballot(valid[r]));
      val[r] = __uint_as_float(order_bits_to_f32_bits((uint32_t)(list[r] >> 32)));
      e_mine[r] = pb.of((double)val[r]);
    }
    bool certified = true;
    double lim = __builtin_inf();
    if (cnt > 0) {
      const uint32_t kk = a.top_k < cnt ? a.top_k : cnt;
      const double tau = (double)__uint_as_float(order_bits_to_f32_bits((uint32_t)(wide_get(list, kk - 1u) >> 32)));
      double ek = 0.0;
#pragma unroll
      for (int r = 0; r < kWideR; ++r) {
        double e = (uint32_t)(r * kWave + lane) < kk ? e_mine[r] : 0.0;
        if (!(e == e)) e = __builtin_inf();
        ek = e > ek ? e : ek;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(ek, off, kWave);
        ek = o > ek ? o : ek;
      }
      lim = tau + ek;
      if (cnt >= a.kp) {
        const double top = (double)__uint_as_float(order_bits_to_f32_bits((uint32_t)(wide_get(list, a.kp - 1u) >> 32)));
        certified = top > lim + pb.global;  // false for NaN / inf
      }
    }
    if (flag0 != 0 || a.force_fail) certified = false;
    // the list is sorted by val and val - e grows with val: the survivors are a prefix, elements 0 .. n_surv - 1
    uint32_t n_surv = 0;
#pragma unroll
    for (int r = 0; r < kWideR; ++r) {
      const bool survivor = certified && valid[r] && !((double)val[r] - e_mine[r] > lim);
      n_surv += (uint32_t)__popcll(__ballot(survivor));
      const uint32_t row = wave_seq_rows(list[r], survivor, lane, pl, pp, a.P, a.list_off);
      s_row[r * kWave + lane] = row;
      s_key[r * kWave + lane] = list[r];
      s_rid[r * kWave + lane] = survivor && row != 0xFFFFFFFFu ? a.row_ids[row] : 0u;
    }
    if (lane == 0) {
      if (a.reset_flag != nullptr) *a.reset_flag = 0u;
      s_failed = certified ? 0u : 1u;
      s_nsurv = n_surv;
      if (!certified) { a.fail_list[atomicAdd(a.fail_list + gridDim.x, 1u)] = q; atomicAdd(a.stats, 1u); }  // fallback_kernel redoes it
    }
  }
  __syncthreads();
  if (s_failed) return;
  const uint32_t n_surv = s_nsurv;
  // ---- exact distances of the survivors: a thread per survivor, the reference's ordered chain over its row (the row-major copy when
  // the index keeps one, else the 16-byte pieces of the lane-transposed tile) against the query in LDS
  uint64_t exact = kKeyMax;
  if (threadIdx.x < n_surv) {
    const uint32_t row = s_row[threadIdx.x];
    const f32x4* xp = a.rows_rm ? reinterpret_cast<const f32x4*>(a.rows_rm + (uint64_t)row * a.ld)
                                : reinterpret_cast<const f32x4*>(a.rows + (uint64_t)(row >> 6) * 64ull * a.ld) + (row & 63);
    const uint64_t xstep = a.rows_rm ? 1 : 64;
    const f32x4* q4p = reinterpret_cast<const f32x4*>(qs);
    float acc = 0.0f;
#pragma unroll 8
    for (uint32_t j = 0; j < a.ld / 4; ++j) {
      const f32x4 x4 = xp[(uint64_t)j * xstep];
      const f32x4 q4 = q4p[j];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (a.metric == 0) {
          const float t = __fsub_rn(x4[c], q4[c]);
          acc = __fadd_rn(acc, __fmul_rn(t, t));
        } else {
          acc = __fadd_rn(acc, __fmul_rn(x4[c], q4[c]));
        }
      }
    }
    if (a.metric) acc = __fsub_rn(1.0f, acc);
    if (acc != acc) atomicOr(a.status, 1u);
    exact = make_key(acc, (uint32_t)s_key[threadIdx.x]);  // (distance, position in the query's probe order)
  }
  __syncthreads();  // every thread has read its approximate key
  s_key[threadIdx.x] = exact;
  __syncthreads();
  // ---- order by the exact key (rank = how many survivors lie below: keys are unique) and emit the top_k
  if (threadIdx.x < n_surv) {
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n_surv; ++j) rank += s_key[j] < exact ? 1u : 0u;
    if (rank < a.top_k) {
      const uint64_t o = (uint64_t)q * a.top_k + rank;
      a.out_ids[o] = s_rid[threadIdx.x];
      a.out_dist[o] = __uint_as_float(order_bits_to_f32_bits((uint32_t)(exact >> 32)));
      if (a.out_keys) a.out_keys[o] = exact;
    }
  }
  const uint32_t n_out = n_surv < a.top_k ? n_surv : a.top_k;
  if (a.out_keys)
    for (uint32_t r = n_out + threadIdx.x; r < a.top_k; r += blockDim.x) a.out_keys[(uint64_t)q * a.top_k + r] = kKeyMax;
  if (threadIdx.x == 0) a.out_count[q] = n_out;
}

}  // namespace vers
