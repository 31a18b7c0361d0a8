// finish.hip.h -- the exact finish of the matrix-core list scan (prescan.hip.h, steps (2) and (3)): ivf_rescore_kernel merges a
// query's partial candidate lists, evaluates the certificate, recomputes the survivors' distances in the reference's own
// arithmetic (ordered f32 chain, base.rs:119-126) and emits the top-k (ivfflat.rs:176-195); fallback_kernel re-scans the
// probed lists of the queries whose certificate failed, exactly.  Included by ivf_search.hip only (the kernels have
// external linkage).
#pragma once
#include "prescan.hip.h"
#include "staged.hip.h"

namespace vers {

// ---- exact finish ---------------------------------------------------------------------------------
struct RescoreArgs {
  const uint64_t* partials;  // [b*P*S_max][kp] approximate keys
  uint32_t P, S_max, kp, top_k, d_pad;
  const uint32_t* pj_list;
  const uint32_t* pj_pref;
  const uint32_t* pj_nq;     // [b*P] partial slots the scan WROTE for (query, probe): the quads of the list, 0 if not scanned here
  const uint32_t* list_off;
  const uint32_t* row_ids;
  const float* rows;
  const float* rows_rm;  // nullable: the same rows row-major [cap_rows][ld] (whole-sector gathers for the exact finish)
  uint32_t ld;
  const float* qp;  // padded queries [b][ldq]
  uint32_t ldq;
  const uint32_t* xmax2_bits;
  const uint32_t* qflags;  // [b*P], slot q*P
  int metric;              // 0 squared L2, 1 cosine distance 1 - dot (the exact chains and the bound follow it)
  int force_fail;          // testing: nothing certifies
  int shadow;              // the vals came from the fp16 shadow: the bound grows by its measured residual (xmax2_bits[2]); 2 = and the query block
                           // was fp16 hi ONLY: the query's own residual |q' - fp16(q')|, summed here next to |q|^2, is charged too (pre_bound)
  uint32_t debug;          // diagnosis (option "scan_debug"): 512 skip the row gather + chains, 1024 merge only 1/8 of the slots
  uint32_t* fail_list;     // [b] out: queries to redo exactly, [b] = their count
  uint32_t* stats;         // [0] += failed queries
  uint32_t* status;
  uint64_t* out_ids;
  float* out_dist;
  uint32_t* out_count;
  uint64_t* out_keys;
  unsigned long long* stamps = nullptr;  // diagnosis (option "scan_debug" & 16): [52..57] cycles of the exact finish's phases, summed over the blocks
  // the flat index's single query (flat_shadow_search1) has no planning kernel to zero its two per-call words: the finish clears the flag it
  // read once it is done with it, the fallback launch the count of queued queries it served (both nullable)
  uint32_t* reset_flag = nullptr;
  uint32_t* reset_count = nullptr;
  // host-pointer single-query call (vers_ivf_search, b == 1, on the shadow): the finish itself stores the stream's status word into the pinned
  // result block behind its results when the certificate held -- the host, spinning on that word, does not wait for the fallback launch that
  // follows and finds nothing to do (3.5-4 us); when the certificate failed the fallback launch publishes, after its re-scan
  const uint32_t* st_word = nullptr;
  uint32_t* st_host = nullptr;
};

// Storage row of the key held by each lane (nprobe mode): seq = position in the query's concatenated probe order,
// pj_pref = first position of probe j.  Whole-wave: lane t holds probe t's (pref, list); per key one ballot finds the
// last scanned probe that starts at or before seq (a per-lane loop over the P probes is 2 P dependent-latency loads).
__device__ __forceinline__ uint32_t wave_seq_rows(uint64_t key, bool valid, int lane, const uint32_t* pj_list, const uint32_t* pj_pref,
                                                  uint32_t P, const uint32_t* list_off) {
  uint32_t my_list = 0xFFFFFFFFu, my_off = 0;
  for (uint32_t c0 = 0; c0 < P; c0 += kWave) {  // 64 probes per chunk (pj_pref ascends with the probe rank: a later match overrides)
    const uint32_t j0 = c0 + (uint32_t)lane;
    const uint32_t pref = j0 < P ? pj_pref[j0] : 0u;
    const uint32_t lst = j0 < P ? pj_list[j0] : 0xFFFFFFFFu;
    uint64_t todo = __ballot(valid);
    while (todo) {
      const int c = __ffsll((unsigned long long)todo) - 1;
      todo &= todo - 1;
      const uint32_t seq = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, c);
      const uint64_t m = __ballot(lst != 0xFFFFFFFFu && pref <= seq);
      if (m == 0) continue;
      const int j = 63 - __builtin_clzll((unsigned long long)m);
      const uint32_t lj = (uint32_t)__builtin_amdgcn_readlane((int)lst, j), pj = (uint32_t)__builtin_amdgcn_readlane((int)pref, j);
      if (lane == c) { my_list = lj; my_off = seq - pj; }
    }
  }
  return valid && my_list != 0xFFFFFFFFu ? list_off[my_list] + my_off : 0xFFFFFFFFu;
}

// The same for P <= 64 with the per-probe operands (and the lists' storage rows, which depend on them) loaded AHEAD of whatever
// produces the keys: two dependent round trips off the tail of a merge.
struct SeqRowsPre { uint32_t pref, lst, loff; };
__device__ __forceinline__ SeqRowsPre wave_seq_rows_load(int lane, const uint32_t* pj_list, const uint32_t* pj_pref, uint32_t P) {
  SeqRowsPre r;
  r.pref = lane < (int)P ? pj_pref[lane] : 0u;
  r.lst = lane < (int)P ? pj_list[lane] : 0xFFFFFFFFu;
  r.loff = 0u;
  return r;
}
__device__ __forceinline__ void wave_seq_rows_load2(SeqRowsPre& r, const uint32_t* list_off) {  // (the dependent half: a round trip later)
  r.loff = r.lst != 0xFFFFFFFFu ? list_off[r.lst] : 0u;
}
__device__ __forceinline__ uint32_t wave_seq_rows_map(uint64_t key, bool valid, int lane, const SeqRowsPre& r) {
  uint32_t row = 0xFFFFFFFFu;
  uint64_t todo = __ballot(valid);
  while (todo) {
    const int c = __ffsll((unsigned long long)todo) - 1;
    todo &= todo - 1;
    const uint32_t seq = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, c);
    const uint64_t m = __ballot(r.lst != 0xFFFFFFFFu && r.pref <= seq);
    if (m == 0) continue;
    const int j = 63 - __builtin_clzll((unsigned long long)m);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)r.loff, j), pj = (uint32_t)__builtin_amdgcn_readlane((int)r.pref, j);
    if (lane == c) row = lo + (seq - pj);
  }
  return row;
}

__device__ __forceinline__ void emit_topk(uint64_t fin, uint32_t q, uint32_t top_k, int lane, const uint32_t* pj_list, const uint32_t* pj_pref,
                                          uint32_t P, const uint32_t* list_off, const uint32_t* row_ids, uint64_t* out_ids, float* out_dist,
                                          uint32_t* out_count, uint64_t* out_keys) {
  const bool have = lane < (int)top_k && fin != kKeyMax;
  const uint64_t o = (uint64_t)q * top_k + lane;
  if (lane < (int)top_k && out_keys) out_keys[o] = have ? fin : kKeyMax;
  const uint32_t row = wave_seq_rows(fin, have, lane, pj_list, pj_pref, P, list_off);
  if (have) {
    out_ids[o] = row_ids[row];
    out_dist[o] = __uint_as_float(order_bits_to_f32_bits((uint32_t)(fin >> 32)));
  }
  const uint64_t hm = __ballot(have);
  if (lane == 0) out_count[q] = (uint32_t)__popcll(hm);
}

// Block of NW waves per query (4: batches; 16: the single query of scan1h_kernel, whose ~250 slots and ~34 survivors are then one
// round of loads and one pass of chains).  All waves merge the partial lists; wave 0 evaluates the certificate; the survivors are dealt
// round-robin to the waves, each of which gathers ITS rows coalesced (the row-major copy: 256 contiguous bytes per quarter
// wave; the tiles: 16-byte pieces), parks the products in its LDS slice and walks the ordered chains over them, one lane per
// survivor (staged.hip.h); wave 0 sorts and emits.  The block is a chain of ~7 dependent memory round trips, so what matters is
// blocks in flight: 20 KB of LDS at d = 768 (the query + 4 x 16 staged product rows of 64 columns) and 4 waves allow the whole
// batch in one round.
// stage_rows == 0 (a query too long for LDS): the chains read HBM directly.
constexpr int kRescoreWaves = 4;
constexpr int kRescoreWaves1 = 16;
inline size_t rescore_lds_bytes(uint32_t ld, bool stage_rows, int nw = kRescoreWaves) {  // the query; then two buffers of 8 staged rows of products per wave
  const size_t exchange = (size_t)nw * kWave * sizeof(uint64_t);  // (the merge's exchange area: inside the product buffers)
  return (size_t)ld * sizeof(float) + (stage_rows ? (size_t)nw * 2 * staged_lds_floats(8) * sizeof(float) : exchange);
}
static_assert(2 * staged_lds_floats(8) * sizeof(float) >= kWave * sizeof(uint64_t), "exchange area");
template <int NW>
__global__ __launch_bounds__(kWave * NW) __attribute__((amdgpu_waves_per_eu(NW == 4 ? 8 : 4, NW == 4 ? 8 : 4))) void ivf_rescore_kernel(RescoreArgs a, int stage_rows) {
  constexpr int kRescoreWaves = NW;  // (shadows the batch width: everything below is per block)
  __shared__ uint32_t srow[kWave];
  __shared__ float sred[kRescoreWaves], sres[kRescoreWaves];
  __shared__ uint32_t s_failed, s_nsurv;
  extern __shared__ __attribute__((aligned(16))) float dyn[];  // the query, padded; then staged rows of pitch ld + 4
  float* qs = dyn;
  float* xs = dyn + a.ld;
  // the merge's exchange area lives in the product buffers (dead until the chains start, a block barrier later): with it the block
  // stays under 40 KB of LDS at d = 768 -- four blocks per CU, the whole batch of 1024 in one round
  uint64_t (*sh)[kWave] = reinterpret_cast<uint64_t(*)[kWave]>(xs);
  const uint32_t q = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned long long ts[5] = {}, tx[6] = {};
  auto stamp = [&](int i) { if (a.stamps) ts[i] = __builtin_amdgcn_s_memtime(); };
  auto xstamp = [&](int i) { if (a.stamps) tx[i] = __builtin_amdgcn_s_memtime(); };
  stamp(0);
  const float* qrow = a.qp + (uint64_t)q * a.ldq;
  const uint32_t* pl = a.pj_list + (uint64_t)q * a.P;
  const uint32_t* pp = a.pj_pref + (uint64_t)q * a.P;
  // everything that does not depend on the merge is requested up front
  const uint32_t flag0 = a.qflags[(uint64_t)q * a.P];
  const uint32_t xmax_bits = *a.xmax2_bits;
  // (the single query's lone block: its slots' keys and minima are requested before anything else -- see the merge below)
  constexpr int kPre = 16;  // keys per thread in registers: 16 k keys = 480 slots of kp = 34
  const uint64_t* keys = a.partials + (uint64_t)q * a.P * a.S_max * a.kp;
  const uint32_t n_slots = a.P * a.S_max;
  const uint64_t n_keys = (uint64_t)n_slots * a.kp;
  const bool flat_ok = NW == kRescoreWaves1 && n_keys <= (uint64_t)8 * kPre * kWave * NW && !(a.debug & 1024u);  // (beyond ~100 k keys -- an index with one very long list -- slot by slot)
  uint64_t pk[kPre];
  uint64_t mn = kKeyMax;
  if constexpr (NW == kRescoreWaves1) {
    if (flat_ok) {
#pragma unroll
      for (int u = 0; u < kPre; ++u) {
        const uint32_t idx = threadIdx.x + (uint32_t)u * (kWave * NW);
        pk[u] = idx < n_keys ? keys[idx] : kKeyMax;
      }
      // (a thread takes two neighbouring slots and keeps the smaller minimum -- more when there are more than 2048 slots --: T stays
      // valid, a little looser, and 448 slots are four waves' worth: two folds instead of three)
      for (uint32_t sl = 2u * threadIdx.x; sl < n_slots; sl += 2u * kWave * NW) {
        const uint64_t k0 = keys[(uint64_t)sl * a.kp], k1 = sl + 1u < n_slots ? keys[(uint64_t)(sl + 1u) * a.kp] : kKeyMax;
        mn = k0 < mn ? k0 : mn;
        mn = k1 < mn ? k1 : mn;
      }
    }
  }
  float qpart = 0.0f, rpart = 0.0f;
  const float qscale = a.metric ? -1.0f : -2.0f;  // (what prescan_kernel_g stages: exact scalings)
  for (uint32_t i = threadIdx.x; i < a.ld; i += blockDim.x) {
    const float v = qrow[i];
    qs[i] = v;
    qpart = __fadd_rn(qpart, __fmul_rn(v, v));  // |q|^2 in any order: the bound inflates it
    if (a.shadow == 2) {  // the staged element's fp16 rounding error, exact in f32 (as x - fp16(x) in shadow_residual_kernel); overflow: inf / NaN
      const float y = qscale * v, dl = y - (float)(_Float16)y;
      rpart = __fadd_rn(rpart, __fmul_rn(dl, dl));
    }
  }
  // wave 0 maps sequence numbers to storage rows twice (survivors, emitted keys): the per-probe operands it needs depend on the
  // query only -- requested here, two dependent round trips off the tail of the merge (P <= 64: a probe per lane)
  const bool pre_ok = a.P <= (uint32_t)kWave;
  SeqRowsPre pre = {0u, 0xFFFFFFFFu, 0u};
  if (wid == 0 && pre_ok) pre = wave_seq_rows_load(lane, pl, pp, a.P);
  // merge the partial lists: wave w folds the live slots w, w + 4, ... ; the four results are folded pairwise.  Only WRITTEN
  // slots are read -- slot (probe j, quad s) exists iff s < pj_nq[j] -- so the slot array needs no 0xFF fill per batch (10 MB
  // at cfg3, and at 8 ranks 7/8 of the slots belong to other GPUs' lists).  A slot is an ascending list (prescan.hip.h,
  // buffer_sorted): two of them merge through a six-stage network (wave_merge_sorted64) -- the ordered inserts of rounds 2-4
  // cost ~130 cycles per key that passed: 26 k of the kernel's 91 k cycles per query at 8 ranks.
  const uint32_t* nqp = a.pj_nq + (uint64_t)q * a.P;
  uint64_t list = kKeyMax;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { qpart += __shfl_xor(qpart, off, kWave); rpart += __shfl_xor(rpart, off, kWave); }
  if (lane == 0) { sred[wid] = qpart; sres[wid] = rpart; }  // (read by wave 0 behind the merge's barriers)
  // The certificate's bound is double-precision arithmetic with four square roots: ~2 us that a batch hides behind its other blocks and
  // the lone block of a single query does not -- there the last wave, idle during the merge, computes it meanwhile.
  __shared__ PreBound s_pb;
  auto make_bound = [&]() {
    float qn = 0.0f, rq = 0.0f;
    for (int w = 0; w < kRescoreWaves; ++w) { qn += sred[w]; rq += sres[w]; }
    return pre_bound((double)qn, (double)__uint_as_float(xmax_bits), a.shadow ? (double)__uint_as_float(a.xmax2_bits[2]) : 0.0, a.d_pad, a.metric, a.shadow, (double)rq);
  };
  bool pb_ready = false;  // (block-uniform)
  if (wid == 0 && pre_ok) wave_seq_rows_load2(pre, a.list_off);  // (the dependent half, in flight under the merge)
  // 64 keys per wave, ascending (kKeyMax padded) -> wave 0 ends with the 64 smallest of the block's, ascending.  Pairwise: in stage s
  // the waves at multiples of 2 s fold in the list of the wave s above (stored a stage ago) and store theirs.
  auto block_fold = [&](uint64_t& v, int nw = NW) {  // nw (a power of two, block-uniform): only the first nw waves hold anything
    sh[wid][lane] = v;
    __syncthreads();
#pragma unroll
    for (int s = 1; s < NW; s <<= 1) {
      if (s >= nw) break;
      if ((wid & (2 * s - 1)) == 0 && wid < nw) {
        wave_merge_sorted64(v, sh[wid + s][lane], lane);
        if (2 * s < nw && wid != 0) sh[wid][lane] = v;
      }
      if (2 * s < nw) __syncthreads();
    }
  };
  bool flat_done = false;  // (block-uniform)
  if constexpr (NW == kRescoreWaves1) {
    // The single query's slots (scan1h_kernel: ~350 at cfg3, 0xFF-filled where nothing was written) as ONE flat array: merging them
    // slot by slot through the six-stage networks was 16 us of this kernel's 29 (a merge is ~0.6 us of dependent LDS-crossbar round
    // trips, a lone block has nothing to hide them behind).  Instead, by COUNTING:
    //   (1) T = the kp-th smallest val among the slots' MINIMA -- at least kp keys lie at or below it, so the kp smallest keys do:
    //       a sort per wave and four folds (counting here -- 448 minima against 448 -- is 200 k compares: 8 us of VALU time);
    //   (2) every thread keeps those of its (already loaded) keys whose val is <= T: ~kp + 2 of them when the rows fell into the
    //       slots at random, the query's whole cluster (~150 at cfg3's 16 modes per list) when its neighbours share a list;
    //   (3) a thread per kept key counts the kept keys below its own: that is its rank -- ranks < 64 are the merged list.
    // More than 1024 kept keys, or a slot area beyond ~100 k keys (an index with one very long list): slot by slot, below.
    constexpr uint32_t kBuf = 1024;
    __shared__ __attribute__((aligned(16))) uint64_t s_buf[kBuf];
    __shared__ uint64_t s_out[kWave];
    __shared__ uint32_t s_rank[kBuf];
    __shared__ uint32_t s_T, s_cnt;
    if (flat_ok) {
      xstamp(0);
      if (threadIdx.x == 0) { s_cnt = 0u; s_T = 0xFFFFFFFFu; }  // (fewer than kp slots hold anything: every key passes)
      if (threadIdx.x < kWave) s_out[threadIdx.x] = kKeyMax;
      s_rank[threadIdx.x] = 0u;
      xstamp(1);
      int nwa = 1;
      while (nwa < NW && (uint32_t)nwa * 2u * kWave < n_slots) nwa *= 2;
      if (wid < nwa) wave_rank_sort64(mn, lane);
      block_fold(mn, nwa);  // (wave 0: the 64 smallest minima, ascending)
      if (wid == 0 && lane == (int)a.kp - 1) s_T = (uint32_t)(mn >> 32);  // (kKeyMax: fewer than kp slots hold anything -- every key passes)
      xstamp(2);
      __syncthreads();
      const uint32_t T = s_T;
      auto keep = [&](uint64_t k) {
        if ((uint32_t)(k >> 32) <= T && k != kKeyMax) {
          const uint32_t pos = atomicAdd(&s_cnt, 1u);
          if (pos < kBuf) s_buf[pos] = k;
        }
      };
#pragma unroll
      for (int u = 0; u < kPre; ++u) keep(pk[u]);
      for (uint64_t i0 = (uint64_t)kPre * kWave * NW; i0 < n_keys; i0 += 4u * kWave * NW) {  // (slots beyond the prefetch: S_max follows the LONGEST list)
        uint64_t k4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const uint64_t idx = i0 + threadIdx.x + (uint64_t)u * (kWave * NW);
          k4[u] = idx < n_keys ? keys[idx] : kKeyMax;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) keep(k4[u]);
      }
      xstamp(3);
      __syncthreads();
      const uint32_t n_in = s_cnt;
      if (n_in <= kBuf) {
        flat_done = true;
        pb_ready = true;
        if (wid == NW - 1) {  // (the last wave has no kept keys to rank unless there are > 960 of them: the certificate's bound meanwhile)
          const PreBound b = make_bound();
          if (lane == 0) s_pb = b;
        }
        // rank of a kept key = the kept keys below it (keys are unique): up to four threads per key, each counting over a quarter
        // of the buffer (broadcast reads: the lanes of a wave hold different keys and walk the same entries)
        const uint32_t parts = n_in == 0 ? 1u : (4u * n_in <= (uint32_t)(kWave * NW) ? 4u : (2u * n_in <= (uint32_t)(kWave * NW) ? 2u : 1u));
        if (threadIdx.x < parts * n_in) {
          const uint32_t ki = threadIdx.x % n_in, part = threadIdx.x / n_in;
          const uint32_t per = (n_in + parts - 1u) / parts, i_begin = part * per, i_end = i_begin + per < n_in ? i_begin + per : n_in;
          const uint64_t my = s_buf[ki];
          uint32_t below = 0;
#pragma unroll 4
          for (uint32_t i = i_begin; i < i_end; ++i) below += s_buf[i] < my ? 1u : 0u;
          if (parts == 1u) s_rank[ki] = below;
          else if (below) atomicAdd(&s_rank[ki], below);
        }
        __syncthreads();
        if (threadIdx.x < n_in) {
          const uint32_t r = s_rank[threadIdx.x];
          if (r < (uint32_t)kWave) s_out[r] = s_buf[threadIdx.x];
        }
        __syncthreads();
        if (wid == 0) list = s_out[lane];
      }
      xstamp(4);
    }
  }
  if (!flat_done) {
    constexpr int U = 4;
    // lane j = probe c0 + j, 64 probes per chunk (nprobe <= 64: one chunk): the slots it wrote, and where they start in the compact
    // order of the chunk's LIVE slots.  At 8 ranks 4 of a query's 32 probes are local: 8 live slots of 64 -- one round of independent
    // loads instead of four rounds of a count load followed by a key load each (20 of the kernel's 84 us there).
    for (uint32_t c0 = 0; c0 < a.P; c0 += kWave) {  // (block-uniform)
      const uint32_t jp = c0 + (uint32_t)lane;
      const uint32_t my_nq = jp < a.P ? (nqp[jp] < a.S_max ? nqp[jp] : a.S_max) : 0u;
      const uint32_t incl = wave_incl_u32(my_nq);
      const uint32_t excl = incl - my_nq;
      uint32_t n_live = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
      if (a.debug & 1024u) n_live = n_live / 8;
      for (uint32_t i0 = 0; (uint32_t)wid + kRescoreWaves * i0 < n_live; i0 += U) {
        uint64_t cand[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t c = (uint32_t)wid + kRescoreWaves * (i0 + u);  // compact index -> (probe j, quad c - first slot of j): the last lane whose slots start at or before c
          const uint64_t m = __ballot(my_nq != 0 && excl <= c);
          const int j = m ? 63 - __builtin_clzll((unsigned long long)m) : 0;
          const uint32_t e_j = (uint32_t)__builtin_amdgcn_readlane((int)excl, j);
          const uint32_t sl = (c0 + (uint32_t)j) * a.S_max + (c - e_j);
          cand[u] = (c < n_live && lane < (int)a.kp) ? keys[(uint64_t)sl * a.kp + lane] : kKeyMax;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) wave_merge_sorted64(list, cand[u], lane);
      }
    }
    block_fold(list);
  }
  stamp(1);
  uint64_t mine = kKeyMax;  // wave 0: the survivors, compacted to lanes 0..n_surv-1
  uint32_t rid = 0;         // wave 0: their vec_ids
  if (wid == 0) {
    if (lane >= (int)a.kp) list = kKeyMax;  // (the networks keep 64 keys: the kp smallest are the candidates)
    const bool valid = lane < (int)a.kp && list != kKeyMax;
    const uint32_t cnt = (uint32_t)__popcll(__ballot(valid));
    const PreBound pb = pb_ready ? s_pb : make_bound();  // (s_pb: written before the merge's last barrier)
    const float val = __uint_as_float(order_bits_to_f32_bits((uint32_t)(list >> 32)));
    const double e_mine = pb.of((double)val);  // this candidate's own bound (NaN / inf vals: NaN / inf, handled by the negated compares)
    bool certified = true;
    double lim = __builtin_inf();
    if (cnt > 0) {
      // With tau the k-th smallest val and e_k the largest bound among the k smallest vals, the k-th smallest D_ref over ALL
      // rows is at most tau + |q|^2 + e_k; a row r can be among the true top-k only if val_r - e_r <= tau + e_k.  Rows in the
      // list: e_r = their own bound (survivors below).  Rows cut off by a full list have val >= the list's last val and the
      // global bound: the last val must clear tau + e_k + E_global.
      const uint32_t kk = a.top_k < cnt ? a.top_k : cnt;
      const double tau = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(val), (int)kk - 1));
      double ek = lane < (int)kk ? e_mine : 0.0;
      if (!(ek == ek)) ek = __builtin_inf();
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(ek, off, kWave);
        ek = o > ek ? o : ek;
      }
      lim = tau + ek;
      if (cnt >= a.kp) {    // a full list may have cut rows off: the kp-th val must clear the limit
        const double top = (double)__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(val), (int)a.kp - 1));
        certified = top > lim + pb.global;  // false for NaN / inf
      }
    }
    if (flag0 != 0 || a.force_fail) certified = false;
    // only candidates inside the limit can reach the top-k: the others are not worth their 3 KiB gather.
    // The list is sorted by val, so the survivors are a prefix: lanes 0..n_surv-1.
    const bool survivor = certified && valid && !((double)val - e_mine > lim);
    const uint32_t n_surv = (uint32_t)__popcll(__ballot(survivor));
    mine = survivor ? list : kKeyMax;
    const uint32_t row = pre_ok ? wave_seq_rows_map(list, survivor, lane, pre) : wave_seq_rows(list, survivor, lane, pl, pp, a.P, a.list_off);
    srow[lane] = row;
    if (survivor && row != 0xFFFFFFFFu) rid = a.row_ids[row];  // (the emitted keys' ids: in flight under the chains)
    if (lane == 0) {
      if (a.reset_flag != nullptr) *a.reset_flag = 0u;  // (read into flag0 at the top; the scan that sets it has finished)
      s_failed = certified ? 0u : 1u;
      s_nsurv = n_surv;
      if (!certified) { a.fail_list[atomicAdd(a.fail_list + gridDim.x, 1u)] = q; atomicAdd(a.stats, 1u); }  // the fallback kernels redo it
    }
  }
  __syncthreads();
  if (s_failed) return;
  stamp(2);
  const uint32_t n_surv = s_nsurv;
  // PUBLISHER INVARIANT of a host-pointer single-query call (a.st_host != nullptr; the host spins on that pinned word and returns the
  // moment it changes, while the fallback launch behind this kernel is still queued): exactly ONE kernel writes st_host per call --
  // this one on every path that ends with its results in place (certified: below and at the end), fallback_kernel when the certificate
  // failed (it returns before touching anything when nothing is queued, so a stale launch never writes under the next call's kStNotYet).
  auto publish_status = [&]() {  // (wave 0: everything it wrote before the status word, at system scope)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane == 0) __hip_atomic_store(a.st_host, __hip_atomic_load(a.st_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  if (a.debug & 512u) {  // (diagnosis: no row gather, no chains -- the approximate order goes out; published like a real result, or every such call would spin its full 2 ms)
    if (wid == 0) {
      emit_topk(list, q, a.top_k, lane, pl, pp, a.P, a.list_off, a.row_ids, a.out_ids, a.out_dist, a.out_count, a.out_keys);
      if (a.st_host != nullptr) publish_status();
    }
    return;
  }
  // exact distances of the survivors: lane per candidate, the reference's ordered chain
  uint64_t cand = kKeyMax;
  bool nan_seen = false;
  const f32x4* q4p = reinterpret_cast<const f32x4*>(qs);
  if (stage_rows) {
    // survivor NW s + w is wave w's staged row s: every wave reads ITS rows coalesced (a quarter wave per row and 64-column
    // chunk), parks the products in its own LDS slice and lane s walks row s's chain (staged.hip.h) -- all survivors in ONE pass
    // over the columns, no block-wide barrier inside.  (Rounds 2-4 staged whole rows 8 at a time for wave 0's lanes: a gather
    // round trip, a barrier and a 3-instruction chain per 8 survivors.)
    __shared__ uint64_t s_cand[kWave];
    // (up to 8 rows per wave and pass -- 32 survivors, in practice all of them: kp <= 34 at top_k = 10 -- so that the kernel
    // stays at 64 registers: beside the two 224-register waves per SIMD of ANOTHER batch's list scan that is what is left, and
    // with batches in flight the finish then runs under that scan instead of waiting for whole CUs)
    constexpr uint32_t kPassRows = 8;
    // The lone block of a single query walks its chains on FOUR of its sixteen waves, one per SIMD: a chain costs its wave ~100
    // VALU instructions per 64 columns whether one lane walks or sixteen do, and with a survivor per wave three chains share each
    // SIMD's issue slots (19 k cycles for 11 survivors; 768 dependent adds are ~7 k).
    constexpr int kCW = NW == kRescoreWaves1 ? 4 : NW;
    if (wid < kCW)
    for (uint32_t base = 0; base < n_surv; base += kPassRows * kCW) {  // block-uniform
      const uint32_t n_pass = n_surv - base < kPassRows * kCW ? n_surv - base : kPassRows * kCW;
      const uint32_t per_wave = (n_pass + kCW - 1) / kCW;
      auto run = [&](auto nl_tag) {
        constexpr int NL = decltype(nl_tag)::value;
        const float* rp[NL];
        const uint32_t xstep = a.rows_rm ? 4u : 256u;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
          const uint32_t sv = base + (uint32_t)(4 * i + (lane >> 4)) * kCW + (uint32_t)wid;
          uint32_t row = srow[sv < n_surv ? sv : 0u];  // (slots past the end re-read survivor 0: never used)
          if (a.debug & 2048u) row &= 1023u;  // (diagnosis: the gather without its address translation / HBM misses)
          rp[i] = (a.rows_rm ? a.rows_rm + (uint64_t)row * a.ld : a.rows + (uint64_t)(row >> 6) * 64ull * a.ld + (row & 63) * 4u) +
                  (uint64_t)(lane & 15) * xstep;
        }
        const float* ql = qs + 4 * (lane & 15);
        constexpr uint32_t kBuf = (uint32_t)staged_lds_floats(kPassRows);
        float* sp = xs + (size_t)wid * 2 * kBuf;
        constexpr int kD = 2;
        return a.metric == 0 ? staged_chains<NL, 0, kD, true, 2, 3>(rp, xstep, ql, a.ld, sp, kBuf, lane)
                             : staged_chains<NL, 1, kD, true, 2, 3>(rp, xstep, ql, a.ld, sp, kBuf, lane);
      };
      float acc = per_wave <= 4 ? run(std::integral_constant<int, 1>{}) : run(std::integral_constant<int, 2>{});
      if (a.metric) acc = __fsub_rn(1.0f, acc);
      const uint32_t sv = base + (uint32_t)lane * kCW + (uint32_t)wid;
      if (lane < (int)kPassRows && sv < n_surv) {
        if (acc != acc) atomicOr(a.status, 1u);
        s_cand[sv] = (uint64_t)f32_to_order_bits(acc) << 32;  // (wave 0 holds the survivors' sequence numbers)
      }
    }
    __syncthreads();
    if (wid == 0 && (uint32_t)lane < n_surv) cand = s_cand[lane] | (uint32_t)mine;
  } else if (wid == 0 && (uint32_t)lane < n_surv) {
    const uint32_t row = srow[lane];
    const f32x4* xp = a.rows_rm ? reinterpret_cast<const f32x4*>(a.rows_rm + (uint64_t)row * a.ld)
                                : reinterpret_cast<const f32x4*>(a.rows + (uint64_t)(row >> 6) * 64ull * a.ld) + (row & 63);
    const uint64_t xstep = a.rows_rm ? 1 : 64;
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
    nan_seen |= acc != acc;
    cand = make_key(acc, (uint32_t)mine);
  }
  if (wid != 0) return;
  stamp(3);
  if (__ballot(nan_seen) != 0 && lane == 0) atomicOr(a.status, 1u);
  uint64_t fin = kKeyMax;
  if constexpr (NW == kRescoreWaves1) {  // (one sort instead of an ordered insert per survivor)
    fin = cand;
    wave_rank_sort64(fin, lane);
    if (lane >= (int)a.top_k) fin = kKeyMax;
  } else {
    wave_topk_update(fin, a.top_k, cand, kKeyMax);
  }
  // emit (emit_topk without its three dependent round trips: the id of an emitted key sits in the lane of the survivor with the
  // same sequence number, loaded before the chains)
  const bool have = lane < (int)a.top_k && fin != kKeyMax;
  const uint64_t o = (uint64_t)q * a.top_k + lane;
  if (lane < (int)a.top_k && a.out_keys) a.out_keys[o] = have ? fin : kKeyMax;
  uint32_t my_id = 0;
  for (uint64_t todo = __ballot(have); todo; todo &= todo - 1) {
    const int t = __ffsll((unsigned long long)todo) - 1;
    const uint32_t seq = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)fin, t);
    const uint64_t m = __ballot((uint32_t)lane < n_surv && (uint32_t)mine == seq);
    const uint32_t idv = (uint32_t)__builtin_amdgcn_readlane((int)rid, m ? __ffsll((unsigned long long)m) - 1 : 0);
    if (lane == t) my_id = idv;
  }
  if (have) {
    a.out_ids[o] = my_id;
    a.out_dist[o] = __uint_as_float(order_bits_to_f32_bits((uint32_t)(fin >> 32)));
  }
  const uint64_t hm = __ballot(have);
  if (lane == 0) a.out_count[q] = (uint32_t)__popcll(hm);
  if (a.st_host != nullptr) publish_status();
  if (a.stamps && lane == 0) {
    stamp(4);
    for (int i = 0; i < 4; ++i) atomicAdd(a.stamps + 52 + i, ts[i + 1] - ts[i]);
    if (tx[4]) { atomicAdd(a.stamps + 58, tx[0] - ts[0]); for (int i = 0; i < 4; ++i) atomicAdd(a.stamps + 59 + i, tx[i + 1] - tx[i]); }
    atomicAdd(a.stamps + 56, 1ull); atomicAdd(a.stamps + 57, (unsigned long long)n_surv);
  }
}

// Exact re-scan of the probed lists of the queries that failed the certificate.  ivf_rescore_kernel queued them.
struct FbSrc {
  static constexpr bool kSeqIds = false;
  static constexpr bool kStreamOnce = true;
  uint64_t* out_ptr;
  uint32_t seq0;
  __device__ __forceinline__ uint32_t seq_base(uint32_t, int) const { return seq0; }
  __device__ __forceinline__ const uint32_t* seq_ids(uint32_t) const { return nullptr; }
  __device__ __forceinline__ uint64_t* out(uint32_t, int) const { return out_ptr; }
  __device__ __forceinline__ uint32_t bound_slot(uint32_t, int) const { return 0; }
};

// ONE launch of kFallbackBlocks persistent blocks; exits at once when nothing is queued (the normal case: ~3 us).
// The blocks split into n_groups groups of G (G = the largest power of two <= blocks / queued queries, at most 64): a group
// takes every n_groups-th queued query and spreads its P probed lists -- in C = ceil(G / P) chunks of tiles each when the
// group is larger than P -- over its members; a member's 8 waves share a chunk's tiles (the ring-pipelined single-query
// item of scan.hip.h) and leave 8 partial lists per chunk in the group's slot.  The member that arrives last (a counter per
// group) folds the slot and emits; the others wait for that before the slot is reused for the group's next query.  With at
// least as many queued queries as blocks G = 1: a block per query, no waiting.  (Round 1 and the first cut of this kernel
// gave a queued query to ONE block: 32 lists = 240 MB at cfg3 through a single CU, ~6 ms -- a cliff behind every failed
// certificate.  Spread over 64 CUs it is ~60 us.)  ctr: [2 * groups + 1] words, zero between launches (the last block out
// clears them).
constexpr uint32_t kFallbackBlocks = 128;
#ifndef VERS_FB_WAVES
#define VERS_FB_WAVES 8  // two waves per SIMD, <= 256 registers: 164, none spilled.  (16 -- the merge kernels' width, rounds 1-5 -- capped the kernel at 128
                         // registers, 95 of them spilled: with EVERY certificate failing 71.6 vs 53.5 ms per 1024 queries at cfg3, same box; the launch
                         // that finds nothing queued costs 3.7-4.1 us either way)
#endif
constexpr int kFbWaves = VERS_FB_WAVES;
inline size_t fallback_part_keys(uint32_t blocks, uint32_t P, uint32_t top_k) {  // >= n_groups * P * C chunks for every G
  return (size_t)blocks * (P > 2 ? P : 2) * kFbWaves * top_k;
}
__global__ __launch_bounds__(kWave * kFbWaves) void fallback_kernel(RescoreArgs a, const uint32_t* list_len, const uint32_t* fail_list,
                                                                       const uint32_t* fail_count, uint64_t* fb_part, uint32_t* ctr,
                                                                       uint32_t* watch, const uint32_t* st_word = nullptr, uint32_t* st_host = nullptr) {
  __shared__ uint64_t sh[kFbWaves][kWave];
  __shared__ uint32_t s_last;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t n_fail = *fail_count;
  // st_host (host-pointer single-query call on the fp16 shadow whose certificate FAILED): the stream's status word goes out to the pinned
  // result block LAST, at system scope, behind the re-scan's results -- the host spins on it (host_io_end)
  auto publish = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(st_host, __hip_atomic_load(st_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  // PUBLISHER INVARIANT (see ivf_rescore_kernel): with nothing queued this launch must not write ANYTHING -- the finish has published the
  // host-pointer call's status, the host may already have returned and started its next call on this workspace (memset of the pinned
  // block, kStNotYet) while this launch is still queued.  Every store below is behind this return.
  if (n_fail == 0) return;
  // `watch` (nullable): pinned host word the host polls to retire an fp16 shadow that fails too often; a.stats[0] is final
  // for this batch (ivf_rescore_kernel, which counts, is done) and only moves when queries were queued
  if (watch != nullptr && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(watch, a.stats[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  uint32_t G = 1;
  // Results wider than one key per lane (the wide finish, finish_wide.hip.h: top_k in (64, 200]) come 64 ranks per PASS here: a pass
  // needs the previous one's last key (ScanParams::lower), which only a block that folds its own slot has at hand -- a block per
  // queued query (G = 1), no cross-block waiting as everywhere else in this kernel.
  const bool wide = a.top_k > (uint32_t)kMaxTopK;
  while (!wide && G < 64u && 2u * G * n_fail <= gridDim.x) G *= 2u;
  const uint32_t n_groups = gridDim.x / G;
  const uint32_t gidx = blockIdx.x / G, g = blockIdx.x % G;
  const uint32_t C = (G + a.P - 1) / a.P, chunks = a.P * C;
  ScanParams p;
  p.ld = a.ld; p.n_chunks = a.ld / kChunk; p.k = a.top_k; p.status = a.status; p.bounds = nullptr; p.lower = nullptr; p.debug = 0;
  p.next_quad = nullptr; p.stamps = nullptr;
  bool nan_seen = false;
  uint64_t* slot = fb_part + (uint64_t)gidx * chunks * kFbWaves * a.top_k;  // the group's chunks x 8 partial lists
  uint32_t* arrived = ctr + 2 * gidx;
  uint32_t round = 0;
  __shared__ uint64_t s_lower;
  if (wide && gidx < n_groups) {  // (G == 1: gidx = blockIdx.x, chunks = P, the slot is this block's own)
    for (uint32_t i = gidx; i < n_fail; i += n_groups) {
      const uint32_t q = fail_list[i];
      const uint32_t* pl = a.pj_list + (uint64_t)q * a.P;
      const uint32_t* pp = a.pj_pref + (uint64_t)q * a.P;
      uint32_t emitted = 0;
      for (uint32_t rank0 = 0; rank0 < a.top_k; rank0 += (uint32_t)kMaxTopK) {  // (block-uniform)
        const uint32_t k_pass = a.top_k - rank0 < (uint32_t)kMaxTopK ? a.top_k - rank0 : (uint32_t)kMaxTopK;
        p.k = k_pass;
        p.lower = rank0 ? &s_lower : nullptr;  // (keys at or below the previous pass's last one are dropped: exactly ranks rank0 .. rank0 + 63 remain)
        for (uint32_t j = 0; j < a.P; ++j) {
          uint64_t* out = slot + ((uint64_t)j * kFbWaves + wid) * k_pass;
          const uint32_t Lj = pl[j];
          uint32_t len = 0, t0 = 0, t1 = 0;
          if (Lj != 0xFFFFFFFFu) {
            len = list_len[Lj];
            const uint32_t n_tiles = (len + kWave - 1) / kWave, per = (n_tiles + kFbWaves - 1) / kFbWaves;
            t0 = (uint32_t)wid * per < n_tiles ? (uint32_t)wid * per : n_tiles;
            t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
          }
          if (t1 <= t0) {
            if (lane < (int)k_pass) out[lane] = kKeyMax;
            continue;
          }
          ItemView<1> v;
          v.rows = a.rows + ((uint64_t)a.list_off[Lj] + (uint64_t)t0 * kWave) * a.ld;
          v.nrows = (t1 * kWave < len ? t1 * kWave : len) - t0 * kWave;
          v.nq = 1;
          v.qb = a.qp + (uint64_t)q * a.ldq;
          FbSrc src;
          src.out_ptr = out;
          src.seq0 = pp[j] + t0 * kWave;
          if (a.metric == 0) scan_item<1, 1, 0>(src, p, 0u, v, lane, nan_seen);
          else scan_item<1, 1, 1>(src, p, 0u, v, lane, nan_seen);
        }
        __threadfence();  // this pass's partial lists before anybody of the block reads them back (the slot's lines may sit in the L1 from the pass before)
        __syncthreads();
        const uint64_t list = block_merge_keys<kFbWaves>(slot, a.P * kFbWaves * k_pass, k_pass, sh);
        if (threadIdx.x < kWave) {
          const bool have = lane < (int)k_pass && list != kKeyMax;
          const uint64_t o = (uint64_t)q * a.top_k + rank0 + (uint32_t)lane;
          if (lane < (int)k_pass && a.out_keys) a.out_keys[o] = have ? list : kKeyMax;
          const uint32_t row = wave_seq_rows(list, have, lane, pl, pp, a.P, a.list_off);
          if (have) {
            a.out_ids[o] = a.row_ids[row];
            a.out_dist[o] = __uint_as_float(order_bits_to_f32_bits((uint32_t)(list >> 32)));
          }
          emitted += (uint32_t)__popcll(__ballot(have));
          if (lane == (int)k_pass - 1) s_lower = list;  // (kKeyMax when the rows ran out: the next pass finds nothing)
        }
        __syncthreads();  // s_lower, and the slot is free for the next pass
      }
      if (threadIdx.x == 0) a.out_count[q] = emitted;
    }
  } else if (gidx < n_groups) {
    for (uint32_t i = gidx; i < n_fail; i += n_groups, ++round) {
      const uint32_t q = fail_list[i];
      for (uint32_t u = g; u < chunks; u += G) {
        const uint32_t j = u / C, c = u % C;
        uint64_t* out = slot + ((uint64_t)u * kFbWaves + wid) * a.top_k;
        const uint32_t Lj = a.pj_list[(uint64_t)q * a.P + j];
        uint32_t len = 0, t0 = 0, t1 = 0;
        if (Lj != 0xFFFFFFFFu) {
          len = list_len[Lj];
          const uint32_t n_tiles = (len + kWave - 1) / kWave, per_c = (n_tiles + C - 1) / C;
          const uint32_t c0 = c * per_c < n_tiles ? c * per_c : n_tiles, c1 = c0 + per_c < n_tiles ? c0 + per_c : n_tiles;
          const uint32_t per = (c1 - c0 + kFbWaves - 1) / kFbWaves;
          t0 = c0 + (uint32_t)wid * per < c1 ? c0 + (uint32_t)wid * per : c1;
          t1 = t0 + per < c1 ? t0 + per : c1;
        }
        if (t1 <= t0) {  // nothing for this wave: an empty slot
          if (lane < (int)a.top_k) out[lane] = kKeyMax;
          continue;
        }
        ItemView<1> v;
        v.rows = a.rows + ((uint64_t)a.list_off[Lj] + (uint64_t)t0 * kWave) * a.ld;
        v.nrows = (t1 * kWave < len ? t1 * kWave : len) - t0 * kWave;
        v.nq = 1;
        v.qb = a.qp + (uint64_t)q * a.ldq;
        FbSrc src;
        src.out_ptr = out;
        src.seq0 = a.pj_pref[(uint64_t)q * a.P + j] + t0 * kWave;
        if (a.metric == 0) scan_item<1, 1, 0>(src, p, 0u, v, lane, nan_seen);
        else scan_item<1, 1, 1>(src, p, 0u, v, lane, nan_seen);
      }
      __threadfence();   // this member's partial lists, device-wide, before it counts as arrived
      __syncthreads();
      bool fold = true;
      if (G > 1) {
        if (threadIdx.x == 0) {
          const uint32_t prev = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
          s_last = prev + 1u == (round + 1u) * G ? 1u : 0u;
        }
        __syncthreads();
        fold = s_last != 0u;
        if (fold) __threadfence();  // the other members' lists
      }
      if (fold) {
        const uint32_t n_keys = chunks * kFbWaves * a.top_k;
        const uint64_t list = block_merge_keys<kFbWaves>(slot, n_keys, a.top_k, sh);
        if (threadIdx.x < kWave)
          emit_topk(list, q, a.top_k, lane, a.pj_list + (uint64_t)q * a.P, a.pj_pref + (uint64_t)q * a.P, a.P, a.list_off, a.row_ids, a.out_ids,
                    a.out_dist, a.out_count, a.out_keys);
        __syncthreads();  // (G == 1: the block's slot is reused by its next query)
      }
      // No block ever WAITS for another one here.  A group's slot would be reused only by a second query of the same group,
      // and a group of G > 1 blocks never has one: G > 1 implies G * n_fail <= gridDim.x (the loop above), i.e.
      // n_groups = gridDim.x / G >= n_fail, so i + n_groups >= n_fail for every i.  (Round 2 carried a spin on a `merged`
      // word for that case -- dead code, but a cross-block spin nonetheless; with G == 1 the block is alone on its slot.)
    }
  }
  if (__ballot(nan_seen) != 0 && lane == 0) atomicOr(a.status, 1u);
  // the last block out leaves the counters zero for the next launch
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t* out_ctr = ctr + 2 * gridDim.x;
    if (__hip_atomic_fetch_add(out_ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x) {
      for (uint32_t w = 0; w <= 2 * gridDim.x; ++w) __hip_atomic_store(ctr + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a.reset_count != nullptr) *a.reset_count = 0u;
      if (st_host != nullptr) publish();  // (every block's emits happened before its arrival at the counter)
    }
  }
}

}  // namespace vers
