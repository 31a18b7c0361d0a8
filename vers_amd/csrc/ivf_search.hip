// ivf_search.hip -- the second half of search_approximate (ivfflat.rs:166-198): the inverted-list scans (single query: item
// records; batches in nprobe mode: matrix-core pre-selection + exact finish, prescan.hip.h; otherwise the ordered-chain
// engine, scan.hip.h), the per-query merges + id mapping, the exhaustive scan of the stored rows (utils.rs:68-82), the
// cross-GPU merge of partial results, host-pointer staging, and the search entry points of the C ABI.
#include <chrono>
#include <cstring>
#include <limits>

#include "finish.hip.h"
#include "flat_shadow.hpp"
#include "ivf_src.hip.h"
#include "single.hip.h"
#include "finish_wide.hip.h"

namespace vers {

// final merge + id mapping: one block per query.  Results wider than 64 keys come 64 ranks per pass (ScanParams::lower):
// this pass emits ranks rank0 .. rank0+63 of every merge group into output row q (pitch top_k) and leaves the group's
// last key as the next pass's lower bound.
struct MergeArgs {
  const uint64_t* partials; uint32_t P, S_max, k_keep; int ref_mode;
  const uint32_t *np, *pj_list, *pj_pref, *pj_take, *list_off, *row_ids;
  uint32_t top_k, rank0;
  uint64_t* out_ids; float* out_dist; uint32_t* out_count; uint64_t* out_keys; uint64_t* lower_out;
  const uint32_t* st_word = nullptr; uint32_t* st_host = nullptr;  // host-pointer single-query call: the stream's status word goes out with the result
};
template <int NW>
__device__ __forceinline__ void ivf_merge_block(const MergeArgs& m, uint32_t q, uint64_t (*sh)[kWave]) {
  const uint32_t P = m.P, S_max = m.S_max, k_keep = m.k_keep, top_k = m.top_k, rank0 = m.rank0;
  const int lane = threadIdx.x & 63;
  const bool w0 = threadIdx.x < kWave;
  const uint64_t* pq = m.partials + (uint64_t)q * P * S_max * k_keep;
  const uint64_t o_base = (uint64_t)q * top_k;
  uint32_t written = 0;
  const uint32_t n_groups = m.ref_mode ? m.np[q] : 1;
  SeqRowsPre pre = {};  // (nprobe mode, P <= 64: what maps a key to its storage row, in flight under the merge)
  const bool pre_ok = !m.ref_mode && P <= (uint32_t)kWave;
  if (w0 && pre_ok) pre = wave_seq_rows_load(lane, m.pj_list + (uint64_t)q * P, m.pj_pref + (uint64_t)q * P, P);
  auto mid = [&]() { if (w0 && pre_ok) wave_seq_rows_load2(pre, m.list_off); };
  if (w0 && m.out_keys && rank0 == 0)
    for (uint32_t i = (uint32_t)lane; i < top_k; i += kWave) m.out_keys[o_base + i] = kKeyMax;  // holes = other GPUs' lists
  for (uint32_t grp = 0; grp < n_groups; ++grp) {
    uint64_t list;
    uint32_t n_emit;
    if (m.ref_mode) {
      const uint32_t take = m.pj_take[(uint64_t)q * P + grp];
      if (take == 0) continue;  // uniform per block
      n_emit = take > rank0 ? (take - rank0 < (uint32_t)kWave ? take - rank0 : (uint32_t)kWave) : 0u;
      if (m.pj_list[(uint64_t)q * P + grp] == kNoList || n_emit == 0) {  // scanned by the GPU that owns the list / this pair is complete
        written += take;
        continue;
      }
      list = block_merge_keys<NW>(pq + (uint64_t)grp * S_max * k_keep, S_max * k_keep, k_keep, sh);
      if (w0) {
        const bool have = lane < (int)n_emit && list != kKeyMax;
        const uint32_t row = have ? m.list_off[m.pj_list[(uint64_t)q * P + grp]] + ((uint32_t)list - m.pj_pref[(uint64_t)q * P + grp]) : 0u;
        if (have) {
          const uint64_t o = o_base + written + rank0 + lane;
          m.out_ids[o] = m.row_ids[row];
          m.out_dist[o] = __uint_as_float(order_bits_to_f32_bits((uint32_t)(list >> 32)));
          if (m.out_keys) m.out_keys[o] = list;
        }
        if (m.lower_out && lane == kWave - 1) m.lower_out[(uint64_t)q * P + grp] = list;
      }
      written += take;
    } else {
      n_emit = top_k - rank0 < (uint32_t)kWave ? top_k - rank0 : (uint32_t)kWave;
      list = block_merge_keys<NW>(pq, P * S_max * k_keep, k_keep, sh, mid);
      if (w0) {
        const bool have = lane < (int)n_emit && list != kKeyMax;
        const uint32_t row = pre_ok ? wave_seq_rows_map(list, have, lane, pre)
                                    : wave_seq_rows(list, have, lane, m.pj_list + (uint64_t)q * P, m.pj_pref + (uint64_t)q * P, P, m.list_off);
        if (have) {
          const uint64_t o = o_base + rank0 + lane;
          m.out_ids[o] = m.row_ids[row];
          m.out_dist[o] = __uint_as_float(order_bits_to_f32_bits((uint32_t)(list >> 32)));
          if (m.out_keys) m.out_keys[o] = list;
        }
        if (m.lower_out && lane == kWave - 1) m.lower_out[(uint64_t)q * P] = list;
        const uint32_t cnt = (uint32_t)__popcll(__ballot(have));
        written = rank0 == 0 || cnt ? rank0 + cnt : 0xFFFFFFFFu;  // (a later pass that finds nothing leaves the count alone)
      }
    }
  }
  if (w0 && lane == 0 && written != 0xFFFFFFFFu && (m.ref_mode ? rank0 == 0 : true)) m.out_count[q] = written;
  // host-pointer single-query call: the status word goes out with the result, LAST and at system scope -- the host spins on this
  // pinned word instead of sleeping in hipStreamSynchronize (host_io_end), so everything wave 0 wrote above must be visible first
  if (m.st_host && w0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane == 0) __hip_atomic_store(m.st_host, __hip_atomic_load(m.st_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
template <int NW>
__global__ __launch_bounds__(kWave * NW) void ivf_merge_kernel(MergeArgs m) {
  __shared__ uint64_t sh[NW][kWave];
  ivf_merge_block<NW>(m, blockIdx.x, sh);
}

// exhaustive merge for the IVF handle (seq == vec_id already): ranks rank0 .. rank0 + k - 1 of output row q (pitch top_k);
// top_k > 64 comes 64 ranks per pass (ScanParams::lower)
__global__ __launch_bounds__(kWave * kMergeWaves) void seg_merge_kernel(const uint64_t* partials, uint32_t n_segs, uint32_t k, uint32_t top_k,
                                                                        uint32_t rank0, uint64_t* out_ids, float* out_dist,
                                                                        uint32_t* out_count, uint64_t* lower_out) {
  __shared__ uint64_t sh[kMergeWaves][kWave];
  const uint32_t q = blockIdx.x;
  uint64_t list = block_merge_keys(partials + (uint64_t)q * n_segs * k, n_segs * k, k, sh);
  if (threadIdx.x >= kWave) return;
  const int lane = threadIdx.x;
  const bool have = lane < (int)k && list != kKeyMax;
  if (have) {
    out_ids[(uint64_t)q * top_k + rank0 + lane] = (uint32_t)list;
    out_dist[(uint64_t)q * top_k + rank0 + lane] = __uint_as_float(order_bits_to_f32_bits((uint32_t)(list >> 32)));
  }
  if (lower_out != nullptr && lane == (int)k - 1) lower_out[q] = list;  // (kKeyMax when the rows ran out: the next pass finds nothing)
  const uint32_t cnt = (uint32_t)__popcll(__ballot(have));
  if (lane == 0 && (rank0 == 0 || cnt)) out_count[q] = rank0 + cnt;
}

// cross-GPU merge of per-rank partial results ([world][b][k] keys + ids, kKeyMax padded): one wave per query.
// nprobe mode: global top-k by key.  reference mode: position p of the output belongs to exactly one rank
// (the owner of the list that position came from), so the merge is a position-wise minimum.
// status != nullptr: a rank whose partial is poisoned (it failed locally, see kStPeerFailed) is reported there.
__global__ __launch_bounds__(kWave) void rank_merge_kernel(const uint64_t* keys, const uint64_t* ids, uint64_t rank_stride,
                                                           uint32_t world, uint32_t b, uint32_t k, int ref_mode,
                                                           uint64_t* out_ids, float* out_dist, uint32_t* out_count, uint32_t* status = nullptr) {
  const uint32_t q = blockIdx.x;
  const int lane = threadIdx.x;
  if (status != nullptr && q == 0)
    for (uint32_t r = lane; r < world; r += kWave)
      if (keys[r * rank_stride] == kKeyMax && ids[r * rank_stride] == kPoisonId) atomicOr(status, kStPeerFailed);
  uint32_t total = 0;
  uint64_t lower = 0;  // nprobe mode, k > 64: 64 ranks per pass, keys at or below the previous pass's last key are skipped
  for (uint32_t r0 = 0; r0 < k; r0 += kWave) {
    const uint32_t kk = k - r0 < (uint32_t)kWave ? k - r0 : (uint32_t)kWave;
    uint64_t key = kKeyMax, id = 0;
    if (ref_mode) {
      if (lane < (int)kk)
        for (uint32_t r = 0; r < world; ++r) {
          const uint64_t kx = keys[r * rank_stride + (uint64_t)q * k + r0 + lane];
          if (kx < key) { key = kx; id = ids[r * rank_stride + (uint64_t)q * k + r0 + lane]; }
        }
    } else {
      uint64_t list = kKeyMax;
      const uint32_t n = world * k;
      for (uint32_t i = 0; i < n; i += kWave) {
        uint64_t cand = kKeyMax;
        if (i + lane < n) cand = keys[(uint64_t)((i + lane) / k) * rank_stride + (uint64_t)q * k + (i + lane) % k];
        if (cand <= lower) cand = kKeyMax;
        wave_topk_update(list, kk, cand, kKeyMax);
      }
      key = lane < (int)kk ? list : kKeyMax;
      if (key != kKeyMax)  // keys are unique: find where this one came from to pick up its id
        for (uint32_t i = 0; i < n; ++i) {
          const uint64_t o = (uint64_t)(i / k) * rank_stride + (uint64_t)q * k + i % k;
          if (keys[o] == key) { id = ids[o]; break; }
        }
      lower = readlane64(list, (int)kk - 1);
    }
    const bool have = key != kKeyMax;
    if (have) {
      out_ids[(uint64_t)q * k + r0 + lane] = id;
      out_dist[(uint64_t)q * k + r0 + lane] = __uint_as_float(order_bits_to_f32_bits((uint32_t)(key >> 32)));
    }
    total += (uint32_t)__popcll(__ballot(have));
    if (!ref_mode && lower == kKeyMax) break;  // fewer keys than ranks: nothing left for later passes
  }
  if (lane == 0) out_count[q] = total;
}

// local exhaustive results -> (key, vec_id) pairs for the cross-GPU merge: key = (order bits of the distance << 32) |
// vec_id, the reference's stable order (utils.rs:77: ties -> lower index); kKeyMax padded
__global__ void pack_exhaustive_keys_kernel(const uint64_t* ids, const float* dist, const uint32_t* cnt, uint32_t b, uint32_t k,
                                            uint64_t* out_keys, uint64_t* out_ids) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b * k) return;
  const uint32_t q = i / k, j = i - q * k;
  const bool have = j < cnt[q];
  out_keys[i] = have ? make_key(dist[i], (uint32_t)ids[i]) : kKeyMax;
  out_ids[i] = have ? ids[i] : ~0ull;
}

}  // namespace vers

namespace vers {
namespace ivf {

template <int QG>
int32_t launch_ivf_scan(vers_ivf* h, const IvfSrc<QG>& src, uint32_t items_bound, hipStream_t st, const uint64_t* lower = nullptr) {
  ScanParams p;
  p.ld = h->ld;
  p.n_chunks = h->ld / kChunk;
  p.k = src.k_keep;
  p.status = W->st_word();
  p.debug = scan_debug_flags();
  p.stamps = nullptr;
  if (p.debug & 16u) {  // diagnosis only
    if (int32_t rc = W->stamps.reserve(512)) return rc;
    VERS_HIP_TRY(hipMemsetAsync(W->stamps.p, 0, 128, st));
    p.stamps = W->stamps.as<unsigned long long>();
  }
  // pruning bounds shared between the items of a merge group (they run at different times here, unlike the flat
  // scans); option "scan_debug" bit 3 switches them off for A/B runs
  p.bounds = (QG != 1 && !(scan_debug_flags() & 8u)) ? W->partials.as<uint64_t>() + W->ivf_bounds_off : nullptr;
  p.lower = lower;
  p.next_quad = nullptr;
  if (QG != 1 && !(scan_debug_flags() & 32u)) {
    if (int32_t rc = W->quad_counter.reserve(16)) return rc;
    VERS_HIP_TRY(hipMemsetAsync(W->quad_counter.p, 0, 16, st));
    p.next_quad = W->quad_counter.as<uint32_t>();
  }
  const size_t lds = scan_lds_bytes(QG, h->ld);
  if (int32_t rc = h->metric ? scan_prepare_launch(scan_kernel<QG, 1, IvfSrc<QG>>, lds) : scan_prepare_launch(scan_kernel<QG, 0, IvfSrc<QG>>, lds)) return rc;
  uint32_t blocks = (items_bound + kWavesPerBlock - 1) / kWavesPerBlock;
  const uint32_t max_blocks = (uint32_t)h->n_cu * scan_blocks_per_cu(QG, h->ld);
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks == 0) blocks = 1;
  const uint32_t slot = (uint32_t)(W->ev_count % SearchWs::kEvRing);
  if (W->ev_on) VERS_HIP_TRY(hipEventRecord(W->ev0[slot], st));
  if (h->metric) hipLaunchKernelGGL((scan_kernel<QG, 1, IvfSrc<QG>>), dim3(blocks), dim3(kWave * kWavesPerBlock), lds, st, src, p);
  else hipLaunchKernelGGL((scan_kernel<QG, 0, IvfSrc<QG>>), dim3(blocks), dim3(kWave * kWavesPerBlock), lds, st, src, p);
  VERS_HIP_TRY(hipGetLastError());
  if (W->ev_on) {
    VERS_HIP_TRY(hipEventRecord(W->ev1[slot], st));
    W->ev_count += 1;
  }
  return VERS_OK;
}

// a single query's list scan over item records (scan1_kernel); timed through the same event ring
int32_t launch_scan1(vers_ivf* h, const Scan1Args& a, uint32_t items_bound, hipStream_t st, const uint64_t* lower, bool few_tiles) {
  ScanParams p;
  p.ld = h->ld;
  p.n_chunks = h->ld / kChunk;
  p.k = a.k_keep;
  p.status = W->st_word();
  p.debug = scan_debug_flags() & ~16u;
  p.stamps = nullptr;
  p.bounds = nullptr;
  p.lower = lower;
  p.next_quad = nullptr;
  uint32_t blocks = (items_bound + kWavesPerBlock - 1) / kWavesPerBlock;  // (W->items holds items_bound + 4 records: one per launched wave)
  const uint32_t max_blocks = (uint32_t)h->n_cu * scan_blocks_per_cu(1, h->ld);
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks == 0) blocks = 1;
  const bool no_ev = !W->ev_on;
  const uint32_t slot = (uint32_t)(W->ev_count % SearchWs::kEvRing);
  // a tile per block of 16 waves (scan1t_kernel: the whole tile in flight at once) instead of a tile per wave; option "scan1t" = 0: the latter
  const bool t1_on = opt_get("scan1t", 1) != 0;
  // ... when the query visits few tiles -- the reference's own mode, a few probes --: with a tile per CU and round, 1361 tiles (nprobe = 32 at
  // cfg3 without a shadow) take 72 us against the tile-per-wave kernel's 58
  if (t1_on && few_tiles && knobs().seg_rows <= 0) {  // (a record is ONE tile unless the tuning knob cut the lists differently)
    if (int32_t rc = h->metric ? scan_prepare_launch(scan1t_kernel<1>, kT1LdsBytes) : scan_prepare_launch(scan1t_kernel<0>, kT1LdsBytes)) return rc;
    const uint32_t t_blocks = std::max<uint32_t>(1u, std::min<uint32_t>(items_bound, (uint32_t)h->n_cu));
    if (!no_ev) VERS_HIP_TRY(hipEventRecord(W->ev0[slot], st));
    if (h->metric) hipLaunchKernelGGL(scan1t_kernel<1>, dim3(t_blocks), dim3(kWave * kT1Waves), kT1LdsBytes, st, a, p);
    else hipLaunchKernelGGL(scan1t_kernel<0>, dim3(t_blocks), dim3(kWave * kT1Waves), kT1LdsBytes, st, a, p);
  } else {
    if (!no_ev) VERS_HIP_TRY(hipEventRecord(W->ev0[slot], st));
    if (h->metric) hipLaunchKernelGGL(scan1_kernel<1>, dim3(blocks), dim3(kWave * kWavesPerBlock), 0, st, a, p);
    else hipLaunchKernelGGL(scan1_kernel<0>, dim3(blocks), dim3(kWave * kWavesPerBlock), 0, st, a, p);
  }
  VERS_HIP_TRY(hipGetLastError());
  if (!no_ev) {
    VERS_HIP_TRY(hipEventRecord(W->ev1[slot], st));
    W->ev_count += 1;
  }
  return VERS_OK;
}

// a single query's list scan on the fp16 shadow (scan1h_kernel: a block per record); timed through the same event ring
int32_t launch_scan1h(vers_ivf* h, const Scan1hArgs& a, uint32_t items_bound, hipStream_t st) {
  const size_t lds = scan1h_lds_bytes(h->ld);
  if (int32_t rc = scan_prepare_launch(scan1h_kernel, lds)) return rc;
  const uint32_t blocks = items_bound ? items_bound : 1u;  // (W->items holds items_bound + 4 records: one per launched block)
  const bool no_ev = !W->ev_on;
  const uint32_t slot = (uint32_t)(W->ev_count % SearchWs::kEvRing);
  if (!no_ev) VERS_HIP_TRY(hipEventRecord(W->ev0[slot], st));
  hipLaunchKernelGGL(scan1h_kernel, dim3(blocks), dim3(kWave * kS1hWaves), lds, st, a);
  VERS_HIP_TRY(hipGetLastError());
  if (!no_ev) {
    VERS_HIP_TRY(hipEventRecord(W->ev1[slot], st));
    W->ev_count += 1;
  }
  return VERS_OK;
}

// the matrix-core list scan (prescan.hip.h); timed through the same event ring as launch_ivf_scan
template <int NQ>
int32_t launch_prescan(vers_ivf* h, const IvfSrc<NQ>& src, uint32_t items_bound, uint32_t kp, uint32_t* qflags, uint32_t* quad_ctr,
                       bool shadow, bool hi_only, hipStream_t st) {
  PreParams p;
  p.rows_bf = shadow ? h->rows_bf.as<uint16_t>() : nullptr;
  p.ld = h->ld;
  p.n_chunks = h->ld / kChunk;
  p.kp = kp;
  p.status = W->st_word();
  p.bounds32 = reinterpret_cast<uint32_t*>(W->partials.as<uint64_t>() + W->ivf_bounds_off);  // 0xFF-initialised with the slots
  p.qflags = qflags;
  p.xnorm = h->xnorm.as<float>();
  p.debug = scan_debug_flags();
  p.metric = (uint32_t)h->metric;
  p.stamps = nullptr;
  if (p.debug & 16u) {
    if (int32_t rc = W->stamps.reserve(512)) return rc;
    VERS_HIP_TRY(hipMemsetAsync(W->stamps.p, 0, 128, st));
    p.stamps = W->stamps.as<unsigned long long>();
  }
  p.next_quad = (p.debug & 32u) ? nullptr : quad_ctr;  // zeroed with the planning tables
  hi_only = hi_only && shadow;
  if (NQ == kPreQWide && !hi_only) return fail(VERS_ERR_INVALID, "internal: 64-query blocks exist on the fp16 shadow with the hi-only query block");
  const size_t lds = prescan_lds_bytes_g(h->ld, kp, NQ, hi_only);
  const bool wide_lists = kp > kPreMaxKp;  // (candidate lists of more than one key per lane: plan_search chose hi-only blocks of 32 or 16 queries on the shadow)
  if (wide_lists && (!hi_only || NQ == kPreQWide || kp > kWideMaxKp)) return fail(VERS_ERR_INVALID, "internal: wide candidate lists need the fp16 shadow with hi-only query blocks of <= 32 queries");
  if constexpr (NQ == kPreQWide) {
    if (int32_t rc = scan_prepare_launch(prescan_kernel_g<true, NQ, IvfSrc<NQ>, false>, lds)) return rc;
  } else if (wide_lists) {
    if (int32_t rc = scan_prepare_launch(prescan_kernel_g<true, NQ, IvfSrc<NQ>, false, true>, lds)) return rc;
  } else {
    if (int32_t rc = hi_only ? scan_prepare_launch(prescan_kernel_g<true, NQ, IvfSrc<NQ>, false>, lds)
                     : shadow ? scan_prepare_launch(prescan_kernel_g<true, NQ, IvfSrc<NQ>>, lds) : scan_prepare_launch(prescan_kernel_g<false, NQ, IvfSrc<NQ>>, lds)) return rc;
  }
  uint32_t blocks = (items_bound + kWavesPerBlock - 1) / kWavesPerBlock;
  uint32_t per_cu = std::max<uint32_t>(1, std::min<uint32_t>(2, (uint32_t)((160u * 1024u) / lds)));  // 1 at d = 768 (measured: as fast as 2)
  const uint32_t reserve = scan_reserve(h, st);  // (vers_set_option "scan_reserve_cus"; auto: only while another batch is in flight)
  const uint32_t max_blocks = ((uint32_t)h->n_cu - reserve) * per_cu;
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks == 0) blocks = 1;
  const uint32_t slot = (uint32_t)(W->ev_count % SearchWs::kEvRing);
  if (W->ev_on) VERS_HIP_TRY(hipEventRecord(W->ev0[slot], st));
  if constexpr (NQ == kPreQWide) {
    hipLaunchKernelGGL((prescan_kernel_g<true, NQ, IvfSrc<NQ>, false>), dim3(blocks), dim3(kWave * kPreWavesG), lds, st, src, p);
  } else if (wide_lists) {
    hipLaunchKernelGGL((prescan_kernel_g<true, NQ, IvfSrc<NQ>, false, true>), dim3(blocks), dim3(kWave * kPreWavesG), lds, st, src, p);
  } else {
    if (hi_only) hipLaunchKernelGGL((prescan_kernel_g<true, NQ, IvfSrc<NQ>, false>), dim3(blocks), dim3(kWave * kPreWavesG), lds, st, src, p);
    else if (shadow) hipLaunchKernelGGL((prescan_kernel_g<true, NQ, IvfSrc<NQ>>), dim3(blocks), dim3(kWave * kPreWavesG), lds, st, src, p);
    else hipLaunchKernelGGL((prescan_kernel_g<false, NQ, IvfSrc<NQ>>), dim3(blocks), dim3(kWave * kPreWavesG), lds, st, src, p);
  }
  VERS_HIP_TRY(hipGetLastError());
  if (W->ev_on) {
    VERS_HIP_TRY(hipEventRecord(W->ev1[slot], st));
    W->ev_count += 1;
  }
  return VERS_OK;
}

// search_approximate for b queries.  nprobe == 0: the reference's own semantics (nearest list,
// spill while short; results concatenated per list).  nprobe >= 1: extension, global top-k
// over the nprobe nearest lists.
int32_t search_dev_locked(vers_ivf* h, const float* q_dev, uint64_t ldq_in, uint32_t b, uint32_t top_k, uint32_t nprobe,
                          uint64_t* out_ids, float* out_dist, uint32_t* out_count, uint64_t* out_keys, hipStream_t st) {
  if (b == 0) return VERS_OK;
  if (top_k == 0) {
    VERS_HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(uint32_t) * b, st));
    return VERS_OK;
  }
  if (h->k == 0) return fail(VERS_ERR_INSUFFICIENT, "search on an index without centroids (reference: index out of bounds, ivfflat.rs:169)");
  // The reference's own mode (nprobe = 0: the nearest list, spilling into the next ones while the result is short, ivfflat.rs:166-195) IS
  // nprobe = 1 whenever no list is shorter than top_k -- nothing can spill: the nearest list's top_k by (distance, list position) either
  // way.  The handle knows its shortest list (lists_that_always_suffice; add() only lengthens lists), so a BATCH in that mode takes
  // the nprobe path: the matrix-core scan of the fp16 shadow + exact finish instead of the ordered chains over the f32 rows (half the
  // bytes: 1.52 -> 0.7 ms per 1024 queries at cfg3).  Single queries keep the f32 tile-per-block scan (scan1t_kernel: one list is
  // latency-bound, the exact finish would cost more than it saves).  option "ref_as_nprobe1" = 0: the ordered chains (A/B runs).
  const bool ref_as_np1 = opt_get("ref_as_nprobe1", 1) != 0;
  if (nprobe == 0 && ref_as_np1 && out_keys == nullptr && h->world == 1 && b >= pre_min_batch_ref().load(std::memory_order_relaxed) && top_k + kPreMinSlack <= kPreMaxKp &&
      knobs().pre_mode != 0 && h->lists_that_always_suffice(top_k) == 1)
    nprobe = 1;
  // Batches below the matrix-core scan's smallest (2 or 3 queries by default) went to one ordered-chain scan of the f32 rows per (query,
  // list) pair: 204 / 272 us at cfg3.  Since round 5 a single query on the shadow is 66 us: such a batch is its queries one after
  // the other on the stream (the per-call tables are reused in stream order, like consecutive calls on one stream).  Same results.
  if (b > 1 && b < pre_min_batch_ref().load(std::memory_order_relaxed) && nprobe != 0 && std::min<uint32_t>(nprobe, h->k) <= (uint32_t)kMaxTopK &&
      top_k + kPreMinSlack <= kPreMaxKp && knobs().pre_mode != 0 && single_shadow_ref().load(std::memory_order_relaxed) != 0 && shadow_mode() != 0 &&
      !h->shadow_off && h->shadow_valid && h->rows_bf.p != nullptr) {
    int32_t rc = VERS_OK;  // (timed -- scan_events 2 -- like single queries: not at all; two event records per query would be 11 us of a batch of 2)
    for (uint32_t qi = 0; qi < b && rc == VERS_OK; ++qi)
      rc = search_dev_locked(h, q_dev + (uint64_t)qi * ldq_in, ldq_in, 1, top_k, nprobe, out_ids + (uint64_t)qi * top_k, out_dist + (uint64_t)qi * top_k,
                             out_count + qi, out_keys ? out_keys + (uint64_t)qi * top_k : nullptr, st);
    return rc;
  }
  SearchPlan s;
  if (int32_t rc = plan_search(h, q_dev, ldq_in, b, top_k, nprobe, st, s)) return rc;
  const uint32_t P = s.P, kp = s.kp, k_keep = s.k_keep, n_pass = s.n_pass, seg_rows = s.seg_rows, seg_target = s.seg_target, S_max = s.S_max;
  const int ref_mode = s.ref_mode, QG = s.QG, pre_mode = s.pre_mode;
  const bool one1 = s.one1, one1_pre = s.one1_pre, use_pre = s.use_pre, use_shadow = s.use_shadow, hi_only = s.pre_hi_only && !s.one1_pre;
  const uint64_t items_bound = s.items_bound;
  const size_t part_bytes = s.part_bytes;
  uint32_t *const pj_list = s.pj_list, *const pj_pref = s.pj_pref, *const pj_take = s.pj_take, *const np = s.np, *const pj_nq = s.pj_nq;
  uint32_t *const cnt = s.cnt, *const pair_off = s.pair_off, *const group_off = s.group_off, *const quad_ctr = s.quad_ctr;
  uint32_t *const fail_list = s.fail_list, *const qflags = s.qflags;
  GroupTotals* const tot = s.tot;
  const float* const qp = s.qp;
  SearchWs::CoarseAhead* const took = s.took;
  auto fill_src = [&](auto& src) {
    src.rows = h->rows.as<float>(); src.ld = h->ld; src.list_off = h->slot_off.as<uint32_t>();  // (items name lists by slot)
    src.list_len = h->slot_len.as<uint32_t>(); src.items = W->items.as<ItemDesc>(); src.n_items_dev = &tot->n_items;
    src.cnt = cnt; src.pair_off = pair_off; src.pairs = W->pairs.as<uint32_t>(); src.group_off = group_off;
    src.qblocks = W->qblocks.as<float>(); src.qp = qp; src.ldq = h->ldq; src.P = P; src.S_max = S_max; src.k_keep = k_keep;
    src.seg_rows = seg_rows; src.seg_target = seg_target; src.pj_pref = pj_pref; src.partials = W->partials.as<uint64_t>();
    src.bound_per_pair = ref_mode ? 1u : 0u;
  };
  int32_t rc;
  if (use_pre) {
    IvfSrc<kPreQ> src; fill_src(src);
    IvfSrc<kPreQNarrow> src_n; fill_src(src_n);  // (rows too long for a 32-query block: 16 queries per block, plan_search chose QG)
    IvfSrc<kPreQWide> src_w; fill_src(src_w);    // (64 queries per block on the shadow, query block as fp16 hi only)
    // partial lists of the exact re-scan (fail_list [b] + its count, qflags [n_pj]: in the zeroed zone above)
    uint32_t fb_blocks = kFallbackBlocks;  // (a power of two; fewer when P x top_k is large: at most 32 MB of partial lists)
    if (b == 1) fb_blocks = 32;  // (a single query: the launch that almost always finds nothing to do costs 4.3 us with 128 blocks of 16 waves to dispatch; a list per block when it does)
    while (fb_blocks > 16 && fallback_part_keys(fb_blocks, P, top_k) * sizeof(uint64_t) > (size_t(32) << 20)) fb_blocks /= 2;
    if (int32_t rc2 = W->fb_part.reserve(fallback_part_keys(fb_blocks, P, top_k) * sizeof(uint64_t))) return rc2;
    if (!W->fb_ctr.p) {  // group counters of fallback_kernel: zero once, the kernel leaves them zero
      if (int32_t rc2 = W->fb_ctr.reserve((2 * kFallbackBlocks + 1) * sizeof(uint32_t))) return rc2;
      VERS_HIP_TRY(hipMemsetAsync(W->fb_ctr.p, 0, (2 * kFallbackBlocks + 1) * sizeof(uint32_t), st));
    }
    if (one1_pre) {  // a single query: its records (plan1_block) against the shadow, a block each
      Scan1hArgs sa;
      sa.rows_h = h->rows_bf.as<uint16_t>(); sa.xnorm = h->xnorm.as<float>(); sa.recs = W->items.as<Item1Rec>(); sa.n_items_dev = &tot->n_items; sa.qp = qp;
      sa.partials = W->partials.as<uint64_t>(); sa.qflags = qflags; sa.ld = h->ld; sa.kp = kp; sa.metric = (uint32_t)h->metric;
      if (int32_t rc2 = launch_scan1h(h, sa, (uint32_t)items_bound, st)) return rc2;
    } else if (int32_t rc2 = QG == kPreQWide     ? launch_prescan(h, src_w, (uint32_t)items_bound, kp, qflags, quad_ctr, use_shadow, hi_only, st)
                             : QG == kPreQNarrow ? launch_prescan(h, src_n, (uint32_t)items_bound, kp, qflags, quad_ctr, use_shadow, hi_only, st)
                                                 : launch_prescan(h, src, (uint32_t)items_bound, kp, qflags, quad_ctr, use_shadow, hi_only, st)) return rc2;
    if (int32_t rc2 = start_pending_ahead(h, st)) return rc2;  // the next batch's coarse quantiser: under this batch's exact finish
    RescoreArgs a;
    a.partials = W->partials.as<uint64_t>(); a.P = P; a.S_max = S_max; a.kp = kp; a.top_k = top_k; a.d_pad = h->ld;
    a.pj_list = pj_list; a.pj_pref = pj_pref; a.pj_nq = pj_nq; a.list_off = h->slot_off.as<uint32_t>(); a.row_ids = h->row_ids.as<uint32_t>();
    a.rows = h->rows.as<float>(); a.rows_rm = h->rows_rm.as<float>(); a.ld = h->ld; a.qp = qp; a.ldq = h->ldq; a.xmax2_bits = h->pre_misc.as<uint32_t>();
    a.qflags = qflags; a.metric = h->metric; a.force_fail = pre_mode == 2; a.shadow = use_shadow ? (hi_only ? 2 : 1) : 0; a.debug = scan_debug_flags(); a.fail_list = fail_list; a.stats = h->pre_misc.as<uint32_t>() + 1;  // (shadow: scan1h_kernel multiplies by the f32 query itself -- no split remainder to charge: 1 covers it)
    a.status = W->st_word(); a.out_ids = out_ids; a.out_dist = out_dist; a.out_count = out_count; a.out_keys = out_keys;
    a.stamps = (scan_debug_flags() & 16u) && W->stamps.p ? W->stamps.as<unsigned long long>() : nullptr;
    if (W->st_host) { a.st_word = W->st_word(); a.st_host = W->st_host; }  // (host-pointer single-query call: the finish publishes when it certifies)
    const int rs_waves = one1_pre ? kRescoreWaves1 : kRescoreWaves;  // (one query: sixteen waves merge its ~250 slots and walk its survivors' chains in one pass)
    const int stage_rows = rescore_lds_bytes(h->ld, true, rs_waves) <= 144u * 1024u ? 1 : 0;
    const size_t rs_lds = rescore_lds_bytes(h->ld, stage_rows != 0, rs_waves);
    if (kp > kPreMaxKp) {  // wide candidate lists (results of 49 .. 200 keys): four keys per lane through the finish (finish_wide.hip.h)
      const size_t w_lds = rescore_wide_lds_bytes(h->ld);
      if (int32_t rc2 = scan_prepare_launch(ivf_rescore_wide_kernel, w_lds)) return rc2;
      hipLaunchKernelGGL(ivf_rescore_wide_kernel, dim3(b), dim3(kWave * kWideWaves), w_lds, st, a);
    } else if (one1_pre) {
      if (int32_t rc2 = scan_prepare_launch(ivf_rescore_kernel<kRescoreWaves1>, rs_lds)) return rc2;
      hipLaunchKernelGGL(ivf_rescore_kernel<kRescoreWaves1>, dim3(b), dim3(kWave * kRescoreWaves1), rs_lds, st, a, stage_rows);
    } else {
      if (int32_t rc2 = scan_prepare_launch(ivf_rescore_kernel<kRescoreWaves>, rs_lds)) return rc2;
      hipLaunchKernelGGL(ivf_rescore_kernel<kRescoreWaves>, dim3(b), dim3(kWave * kRescoreWaves), rs_lds, st, a, stage_rows);
    }
    VERS_HIP_TRY(hipGetLastError());
    if (W->ev_on) {  // (measurement hook vers_ivf_last_finish_ms: from the scan's end record to here)
      VERS_HIP_TRY(hipEventRecord(W->evf, st));
      W->evf_valid = true;
      W->evf_slot = (uint32_t)((W->ev_count - 1) % SearchWs::kEvRing);
    }
    hipLaunchKernelGGL(fallback_kernel, dim3(fb_blocks), dim3(kWave * kFbWaves), 0, st, a, (const uint32_t*)h->slot_len.as<uint32_t>(),
                       (const uint32_t*)fail_list, (const uint32_t*)(fail_list + b), W->fb_part.as<uint64_t>(), W->fb_ctr.as<uint32_t>(),
                       use_shadow ? h->fail_watch : (uint32_t*)nullptr, (const uint32_t*)W->st_word(), W->st_host);  // (st_host: host-pointer single-query call)
    VERS_HIP_TRY(hipGetLastError());
    W->last_pre.valid = true; W->last_pre.b = b; W->last_pre.P = P; W->last_pre.S_max = S_max; W->last_pre.kp = kp; W->last_pre.top_k = top_k;
    W->last_pre.qp = qp; W->last_pre.shadow = use_shadow ? (hi_only ? 2 : 1) : 0;
    if (use_shadow) h->shadow_queries += b;  // (fallback_kernel writes the running failure count to the pinned watch word)
    h->pre_batches += 1;
    W->tot_valid = true;
    if (took) { VERS_HIP_TRY(hipEventRecord(took->freed, st)); took->freed_rec = true; }
    return VERS_OK;
  }
  // ordered-chain scans: 64 result ranks per pass (one pass for top_k <= 64)
  for (uint32_t pass = 0; pass < n_pass; ++pass) {
    const uint64_t* lower = pass ? W->lower.as<uint64_t>() : nullptr;
    if (pass) VERS_HIP_TRY(hipMemsetAsync(W->partials.p, 0xFF, part_bytes, st));  // slots and pruning bounds of the previous pass
    MergeArgs ma;
    ma.partials = W->partials.as<uint64_t>(); ma.P = P; ma.S_max = S_max; ma.k_keep = k_keep; ma.ref_mode = ref_mode; ma.np = np;
    ma.pj_list = pj_list; ma.pj_pref = pj_pref; ma.pj_take = pj_take; ma.list_off = h->slot_off.as<uint32_t>(); ma.row_ids = h->row_ids.as<uint32_t>();
    ma.top_k = top_k; ma.rank0 = pass * (uint32_t)kMaxTopK; ma.out_ids = out_ids; ma.out_dist = out_dist; ma.out_count = out_count; ma.out_keys = out_keys;
    ma.lower_out = n_pass > 1 ? W->lower.as<uint64_t>() : (uint64_t*)nullptr;
    if (W->st_host && pass + 1 == n_pass) { ma.st_word = W->st_word(); ma.st_host = W->st_host; }
    if (one1) {  // the single query's items are records (plan1_block)
      Scan1Args sa;
      sa.rows = h->rows.as<float>(); sa.recs = W->items.as<Item1Rec>(); sa.n_items_dev = &tot->n_items; sa.qp = qp;
      sa.partials = W->partials.as<uint64_t>(); sa.k_keep = k_keep; sa.S_max = S_max; sa.bound_per_pair = ref_mode ? 1u : 0u;
      const bool few_tiles = ref_mode || (uint64_t)P * (h->n_total / std::max<uint32_t>(1, h->k)) / kWave <= 2ull * (uint64_t)h->n_cu;
      rc = launch_scan1(h, sa, (uint32_t)items_bound, st, lower, few_tiles);
    } else if (QG == 1) {
      IvfSrc<1> src; fill_src(src);
      rc = launch_ivf_scan(h, src, (uint32_t)items_bound, st, lower);
    } else if (QG == 8) {
      IvfSrc<8> src; fill_src(src);
      rc = launch_ivf_scan(h, src, (uint32_t)items_bound, st, lower);
    } else {
      IvfSrc<16> src; fill_src(src);
      rc = launch_ivf_scan(h, src, (uint32_t)items_bound, st, lower);
    }
    if (rc) return rc;
    if (pass == 0)
      if (int32_t rc2 = start_pending_ahead(h, st)) return rc2;
    // four waves while a wave can hold its share of the slots' heads in registers (4096 slots of a merge group: two tree levels
    // instead of four; same box, single query: 94.2 -> 91.6 us, reference mode 59.0 -> 55.3), sixteen beyond
    const uint64_t slots_per_group = ref_mode ? S_max : (uint64_t)P * S_max;
    if (slots_per_group <= 4096) hipLaunchKernelGGL(ivf_merge_kernel<4>, dim3(b), dim3(kWave * 4), 0, st, ma);
    else hipLaunchKernelGGL(ivf_merge_kernel<kMergeWaves>, dim3(b), dim3(kWave * kMergeWaves), 0, st, ma);
    VERS_HIP_TRY(hipGetLastError());
  }
  W->tot_valid = true;
  if (took) { VERS_HIP_TRY(hipEventRecord(took->freed, st)); took->freed_rec = true; }
  return VERS_OK;
}

int32_t exhaustive_dev_locked(vers_ivf* h, const float* q_dev, uint64_t ldq_in, uint32_t b, uint32_t top_k, uint32_t metric,
                              uint64_t* out_ids, float* out_dist, uint32_t* out_count, hipStream_t st) {
  if (b == 0) return VERS_OK;
  if (top_k == 0) {
    VERS_HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(uint32_t) * b, st));
    return VERS_OK;
  }
  const int QG = b == 1 ? 1 : 8;
  const uint32_t n_qg = (b + QG - 1) / QG;
  if (int32_t rc = W->qil.reserve((size_t)n_qg * h->ldq * QG * sizeof(float))) return rc;
  if (int32_t rc = launch_stage_queries(q_dev, ldq_in, h->d, W->qil.as<float>(), h->ldq, b, QG, st)) return rc;
  const uint32_t target_items = (uint32_t)h->n_cu * scan_blocks_per_cu(QG, h->ld) * kWavesPerBlock;
  uint64_t per = (h->cap_rows * n_qg + target_items - 1) / target_items;
  const uint32_t seg_rows = (uint32_t)std::min<uint64_t>(round_up64(per ? per : 1, kWave), max_seg_rows(h->ld));
  uint32_t n_segs = (uint32_t)((h->cap_rows + seg_rows - 1) / seg_rows);
  if (n_segs == 0) n_segs = 1;
  const uint32_t k_w = std::min<uint32_t>(top_k, kMaxTopK);  // one key per lane; wider results: 64 ranks per pass (utils.rs:68-82 has no cap)
  if (int32_t rc = W->xpart.reserve((size_t)b * n_segs * k_w * sizeof(uint64_t))) return rc;
  if (top_k > (uint32_t)kMaxTopK)
    if (int32_t rc = W->lower.reserve((size_t)b * sizeof(uint64_t))) return rc;
  const uint32_t n_segs_pad = QG == 1 ? n_segs : round_up(n_segs, 4);
  // (a pass past the last stored row finds nothing; 64-bit rank: no wrap near 2^32)
  for (uint64_t rank0 = 0; rank0 < top_k && (rank0 == 0 || rank0 < h->cap_rows); rank0 += kMaxTopK) {
    const uint32_t k_pass = (uint32_t)std::min<uint64_t>(kMaxTopK, top_k - rank0);
    const uint64_t* lower = rank0 ? W->lower.as<uint64_t>() : nullptr;
    auto fill = [&](auto& src) {
      src.rows = h->rows.as<float>(); src.n = h->cap_rows; src.ld = h->ld; src.seg_rows = seg_rows; src.n_segs = n_segs;
      src.n_segs_pad = n_segs_pad;
      src.queries = W->qil.as<float>(); src.ldq = h->ldq; src.b = b; src.partials = W->xpart.as<uint64_t>(); src.k = k_pass;
      src.ids = h->row_ids.as<uint32_t>();
    };
    int32_t rc;
    if (QG == 1) {
      SegSrc<1, true> src; fill(src);
      rc = launch_seg_scan(h, src, n_segs_pad * n_qg, (int)metric, st, lower);
    } else {
      SegSrc<8, true> src; fill(src);
      rc = launch_seg_scan(h, src, n_segs_pad * n_qg, (int)metric, st, lower);
    }
    if (rc) return rc;
    hipLaunchKernelGGL(seg_merge_kernel, dim3(b), dim3(kWave * kMergeWaves), 0, st, W->xpart.as<uint64_t>(), n_segs, k_pass, top_k, (uint32_t)rank0, out_ids,
                       out_dist, out_count, top_k > (uint32_t)kMaxTopK ? W->lower.as<uint64_t>() : (uint64_t*)nullptr);
    VERS_HIP_TRY(hipGetLastError());
  }
  return VERS_OK;
}

int32_t ensure_out(vers_ivf* h, size_t need, uint32_t b) {
  if (int32_t rc = W->o_ids.reserve(need * sizeof(uint64_t))) return rc;
  if (int32_t rc = W->o_dist.reserve(need * sizeof(float))) return rc;
  return W->o_cnt.reserve((size_t)b * sizeof(uint32_t));
}

int32_t upload_queries(const float* queries, uint64_t stride_bytes, uint32_t b, uint32_t d, DevBuf& buf) {
  if (int32_t rc = buf.reserve((size_t)b * d * sizeof(float))) return rc;
  VERS_HIP_TRY(hipMemcpy2D(buf.p, (size_t)d * 4, queries, stride_bytes, (size_t)d * 4, b, hipMemcpyHostToDevice));
  return VERS_OK;
}

int32_t download_results(vers_ivf* h, uint32_t b, uint32_t top_k, uint64_t* out_ids, float* out_dist, uint32_t* out_count) {
  const size_t need = (size_t)b * top_k;
  if (need) {
    VERS_HIP_TRY(hipMemcpy(out_ids, W->o_ids.p, need * sizeof(uint64_t), hipMemcpyDeviceToHost));
    VERS_HIP_TRY(hipMemcpy(out_dist, W->o_dist.p, need * sizeof(float), hipMemcpyDeviceToHost));
  }
  VERS_HIP_TRY(hipMemcpy(out_count, W->o_cnt.p, (size_t)b * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return VERS_OK;
}

// ---- host-pointer calls: staged through pinned memory, one synchronisation -----------------------------
struct HostIo {
  bool direct = false;  // single query: the kernels write ids / distances / count / status straight into the pinned block
  size_t q_bytes, ids_off, dist_off, cnt_off, st_off, out_bytes;
  float* q_dev;
  uint64_t* ids_dev;
  float* dist_dev;
  uint32_t* cnt_dev;
};

int32_t host_io_begin(vers_ivf* h, const float* queries, uint64_t stride_bytes, uint32_t b, uint32_t top_k, HostIo& io, bool direct = false) {
  const size_t need = (size_t)b * std::max<uint32_t>(top_k, 1);
  io.q_bytes = (size_t)b * h->d * sizeof(float);
  io.ids_off = 0;
  io.dist_off = need * sizeof(uint64_t);
  io.cnt_off = io.dist_off + need * sizeof(float);
  io.st_off = io.cnt_off + (size_t)b * sizeof(uint32_t);
  io.out_bytes = io.st_off + 16;
  if (!W->io_stream) VERS_HIP_TRY(hipStreamCreateWithFlags(&W->io_stream, hipStreamNonBlocking));
  if (W->used && W->last_stream != W->io_stream) VERS_HIP_TRY(hipStreamWaitEvent(W->io_stream, W->done, 0));
  if (int32_t rc = W->io_q.reserve(io.q_bytes)) return rc;
  if (int32_t rc = W->io_out.reserve(io.out_bytes)) return rc;
  // (a single query's results are written into the pinned block by the merge launch itself -- no memset, no device-to-device
  // and device-to-host copies behind the search: three stream operations of ~4 us each, 124 -> 112 us per call; the query's bytes
  // and the results do not share pinned bytes then.  Letting coarse1_kernel read the QUERY from the pinned block too instead of
  // the H2D copy measured the same: not kept)
  io.direct = direct;
  const size_t q_pin = (io.q_bytes + 63) & ~(size_t)63;
  const size_t pin_need = io.direct ? q_pin + io.out_bytes : std::max(io.q_bytes, io.out_bytes);
  if (pin_need > W->io_pin_cap) {
    if (W->io_pin) (void)hipHostFree(W->io_pin);
    W->io_pin = nullptr; W->io_pin_cap = 0;
    VERS_HIP_TRY(hipHostMalloc(&W->io_pin, pin_need, hipHostMallocDefault));
    W->io_pin_cap = pin_need;
  }
  char* base = io.direct ? (char*)W->io_pin + q_pin : (char*)W->io_out.p;
  io.q_dev = W->io_q.as<float>();
  io.ids_dev = (uint64_t*)(base + io.ids_off);
  io.dist_dev = (float*)(base + io.dist_off);
  io.cnt_dev = (uint32_t*)(base + io.cnt_off);
  for (uint32_t i = 0; i < b; ++i)
    std::memcpy((char*)W->io_pin + (size_t)i * h->d * 4, (const char*)queries + (size_t)i * stride_bytes, (size_t)h->d * 4);
  VERS_HIP_TRY(hipMemcpyAsync(io.q_dev, W->io_pin, io.q_bytes, hipMemcpyHostToDevice, W->io_stream));
  return VERS_OK;
}

// copies results + status word back, waits once, maps the status; *out_status_rc carries kRetrySpill etc.
int32_t host_io_end(vers_ivf* h, const HostIo& io, uint32_t b, uint32_t top_k, uint64_t* out_ids, float* out_dist, uint32_t* out_count) {
  const char* pin = (const char*)W->io_pin;
  if (io.direct) {
    pin = (const char*)io.ids_dev - io.ids_off;
    // The result block is pinned and its status word is the call's LAST store (system scope, behind a release fence): spin on it for
    // up to 2 ms -- a single query takes ~90 us -- instead of sleeping in hipStreamSynchronize, whose wake-up costs the call several
    // microseconds; a word that stays "not yet" (an error path that launched no merge, a very slow call) falls back to the synchronisation.
    const volatile uint32_t* flag = reinterpret_cast<const volatile uint32_t*>(pin + io.st_off);
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (uint32_t spins = 0; host_spin_ref().load(std::memory_order_relaxed) != 0; ++spins) {
      if (*flag != kStNotYet) { seen = true; break; }
      if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
      __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen) VERS_HIP_TRY(hipStreamSynchronize(W->io_stream));
  } else {
    char* base = (char*)W->io_out.p;
    VERS_HIP_TRY(hipMemcpyAsync(base + io.st_off, W->st_word(), sizeof(uint32_t), hipMemcpyDeviceToDevice, W->io_stream));
    VERS_HIP_TRY(hipMemcpyAsync(W->io_pin, base, io.out_bytes, hipMemcpyDeviceToHost, W->io_stream));
    VERS_HIP_TRY(hipStreamSynchronize(W->io_stream));
  }
  uint32_t s = 0;
  std::memcpy(&s, pin + io.st_off, sizeof(s));
  if (int32_t rc = status_to_rc(h, s, W->st_slot)) return rc;
  const size_t need = (size_t)b * top_k;
  if (need) {
    std::memcpy(out_ids, pin + io.ids_off, need * sizeof(uint64_t));
    std::memcpy(out_dist, pin + io.dist_off, need * sizeof(float));
  }
  std::memcpy(out_count, pin + io.cnt_off, (size_t)b * sizeof(uint32_t));
  return VERS_OK;
}

}  // namespace ivf
}  // namespace vers

namespace vers {

// ---- the flat index's shadow (flat_shadow.hpp): derive, gate, search ------------------------------------------------------------
namespace {
__global__ void flat_row_ids_kernel(uint32_t* ids, uint64_t n, uint64_t n_pad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pad) ids[i] = i < n ? (uint32_t)i : 0xFFFFFFFFu;
}
constexpr uint32_t kFsTables = 8;  // misc[8 ..]: pj_list | pj_pref | pj_nq | list_off | list_len | qflags | fail_list | fail count
constexpr uint32_t kFsSlotKeys = 64;
}  // namespace

void FlatShadow::release() {
  for (void* p : {(void*)rows_h, (void*)xnorm, (void*)row_ids, (void*)misc, (void*)slots, (void*)fb_part, (void*)fb_ctr})
    if (p) (void)hipFree(p);
  if (bytes) dev_mem_account(-(int64_t)bytes);
  *this = FlatShadow();
}

int32_t flat_shadow_derive(FlatShadow& s, const float* rows_blocked, uint64_t n, uint32_t ld, int n_cu) {
  s.release();
  if (n == 0 || shadow_mode() == 0) return VERS_OK;
  const uint64_t n_pad = (n + 63) / 64 * 64;
  const size_t b_rows = n_pad * (size_t)ld * sizeof(uint16_t), b_norm = n_pad * sizeof(float), b_ids = n_pad * sizeof(uint32_t);
  if (b_rows >= (size_t(1) << 32)) return VERS_OK;  // (flat1h_kernel addresses the shadow through one buffer descriptor)
  s.n_slots = 2u * (uint32_t)n_cu;
  const uint32_t fb_blocks = kFallbackBlocks;
  const size_t b_slots = (size_t)s.n_slots * kFsSlotKeys * sizeof(uint64_t), b_fb = fallback_part_keys(fb_blocks, 1, kMaxTopK) * sizeof(uint64_t),
               b_ctr = (2 * kFallbackBlocks + 1) * sizeof(uint32_t);
  // optional memory: without it the f32 ordered-chain scan stays in charge
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  const size_t need = b_rows + b_norm + b_ids + b_slots + b_fb + b_ctr + 256;
  bool ok = need + (size_t(1) << 30) <= free_b;
  ok = ok && hipMalloc((void**)&s.rows_h, b_rows) == hipSuccess && hipMalloc((void**)&s.xnorm, b_norm) == hipSuccess &&
       hipMalloc((void**)&s.row_ids, b_ids) == hipSuccess && hipMalloc((void**)&s.misc, 256) == hipSuccess &&
       hipMalloc((void**)&s.slots, b_slots) == hipSuccess && hipMalloc((void**)&s.fb_part, b_fb) == hipSuccess &&
       hipMalloc((void**)&s.fb_ctr, b_ctr) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    s.release();
    return VERS_OK;
  }
  s.bytes = need;
  dev_mem_account((int64_t)s.bytes);
  uint32_t host_misc[64] = {};
  host_misc[kFsTables + 2] = s.n_slots;  // pj_nq: every slot is written by every search
  host_misc[kFsTables + 4] = (uint32_t)n;  // list_len
  VERS_HIP_TRY(hipMemcpy(s.misc, host_misc, sizeof(host_misc), hipMemcpyHostToDevice));
  VERS_HIP_TRY(hipMemset(s.fb_ctr, 0, b_ctr));
  VERS_HIP_TRY(hipMemset(s.slots, 0xFF, b_slots));
  hipLaunchKernelGGL(flat_row_ids_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, 0, s.row_ids, n, n_pad);
  const uint64_t work = n_pad * (ld / 8);
  hipLaunchKernelGGL(rows_to_f16_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, 0, rows_blocked, ld, (uint64_t)0, n_pad, s.rows_h);
  hipLaunchKernelGGL(shadow_residual_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, 0, rows_blocked, ld, (const uint32_t*)s.row_ids, (uint64_t)0, n_pad, s.misc + 2);
  hipLaunchKernelGGL(blocked_row_norms_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, 0, rows_blocked, ld, (const uint32_t*)s.row_ids, (uint64_t)0, n_pad, s.xnorm, s.misc);
  VERS_HIP_TRY(hipGetLastError());
  VERS_HIP_TRY(hipDeviceSynchronize());
  uint32_t r2 = 0;
  VERS_HIP_TRY(hipMemcpy(&r2, s.misc + 2, sizeof(r2), hipMemcpyDeviceToHost));
  float r2f;
  std::memcpy(&r2f, &r2, sizeof(r2f));
  if (!(r2f < std::numeric_limits<float>::infinity())) {  // an element overflows fp16: no certificate would hold
    s.release();
    return VERS_OK;
  }
  s.rows_built = n;
  return VERS_OK;
}

bool flat_shadow_usable(const FlatShadow& s, uint64_t n, uint32_t ld, uint32_t top_k) {
  return s.rows_built == n && n != 0 && s.rows_h != nullptr && shadow_mode() != 0 && single_shadow_ref().load(std::memory_order_relaxed) != 0 &&
         knobs().pre_mode != 0 && top_k >= 1 && top_k + kPreMinSlack <= kPreMaxKp && scan1h_lds_bytes(ld) <= 64u * 1024u;
}

int32_t flat_shadow_search1(FlatShadow& s, const float* rows_blocked, uint64_t n, uint32_t ld, int n_cu, const float* q_padded, uint32_t top_k,
                            uint32_t metric, uint32_t* status, uint64_t* out_ids, float* out_dist, uint32_t* out_count, hipStream_t st,
                            hipEvent_t ev0, hipEvent_t ev1) {
  const uint32_t kp = std::min<uint32_t>(kPreMaxKp, top_k + std::max<uint32_t>(24, top_k));  // (the inverted lists' slack on the shadow: ivf_plan.hip)
  uint32_t* const tb = s.misc + kFsTables;
  const uint32_t n_tiles = (uint32_t)((n + 63) / 64);
  const uint32_t blocks = std::min<uint32_t>(s.n_slots, (n_tiles + kS1hWaves - 1) / kS1hWaves);
  Flat1hArgs fa;
  fa.rows_h = s.rows_h; fa.xnorm = s.xnorm; fa.qp = q_padded; fa.partials = s.slots; fa.qflags = tb + 5; fa.ld = ld; fa.kp = kp; fa.metric = metric; fa.n_rows = (uint32_t)n;
  const size_t lds = scan1h_lds_bytes(ld);
  if (int32_t rc = scan_prepare_launch(flat1h_kernel, lds)) return rc;
  if (blocks < s.n_slots)  // (a small corpus: the slots no block writes must read as empty; kp may differ from the last call's)
    VERS_HIP_TRY(hipMemsetAsync(s.slots, 0xFF, (size_t)s.n_slots * kFsSlotKeys * sizeof(uint64_t), st));
  if (ev0) VERS_HIP_TRY(hipEventRecord(ev0, st));
  hipLaunchKernelGGL(flat1h_kernel, dim3(blocks), dim3(kWave * kS1hWaves), lds, st, fa);
  VERS_HIP_TRY(hipGetLastError());
  if (ev1) VERS_HIP_TRY(hipEventRecord(ev1, st));
  RescoreArgs a;
  a.partials = s.slots; a.P = 1; a.S_max = s.n_slots; a.kp = kp; a.top_k = top_k; a.d_pad = ld;
  a.pj_list = tb + 0; a.pj_pref = tb + 1; a.pj_nq = tb + 2; a.list_off = tb + 3; a.row_ids = s.row_ids;
  a.rows = rows_blocked; a.rows_rm = nullptr; a.ld = ld; a.qp = q_padded; a.ldq = ld; a.xmax2_bits = s.misc;
  a.qflags = tb + 5; a.metric = (int)metric; a.force_fail = knobs().pre_mode == 2; a.shadow = 1; a.debug = 0; a.fail_list = tb + 6; a.stats = s.misc + 1;
  a.status = status; a.out_ids = out_ids; a.out_dist = out_dist; a.out_count = out_count; a.out_keys = nullptr; a.stamps = nullptr;
  a.reset_flag = tb + 5; a.reset_count = tb + 7;
  const int stage_rows = rescore_lds_bytes(ld, true, kRescoreWaves1) <= 144u * 1024u ? 1 : 0;
  const size_t rs_lds = rescore_lds_bytes(ld, stage_rows != 0, kRescoreWaves1);
  if (int32_t rc = scan_prepare_launch(ivf_rescore_kernel<kRescoreWaves1>, rs_lds)) return rc;
  hipLaunchKernelGGL(ivf_rescore_kernel<kRescoreWaves1>, dim3(1), dim3(kWave * kRescoreWaves1), rs_lds, st, a, stage_rows);
  VERS_HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(fallback_kernel, dim3(kFallbackBlocks), dim3(kWave * kFbWaves), 0, st, a, (const uint32_t*)(tb + 4), (const uint32_t*)(tb + 6),
                     (const uint32_t*)(tb + 7), s.fb_part, s.fb_ctr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr);
  VERS_HIP_TRY(hipGetLastError());
  return VERS_OK;
}

}  // namespace vers

extern "C" {

int32_t vers_ivf_search_dev(vers_ivf_t* h, const float* queries_dev, uint64_t ldq_floats, uint32_t b, uint32_t top_k,
                            uint32_t nprobe, uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_count_dev, void* stream) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  if (b && (!queries_dev || ldq_floats < h->d || !out_count_dev || (top_k && (!out_ids_dev || !out_dist_dev))))
    return fail(VERS_ERR_INVALID, "vers_ivf_search_dev: bad arguments");
  std::shared_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  WsLease lease(h, true, (hipStream_t)stream);
  if (lease.rc) return lease.rc;
  if (int32_t rc = lease.order_on((hipStream_t)stream)) return rc;
  return search_dev_locked(h, queries_dev, ldq_floats, b, top_k, nprobe, out_ids_dev, out_dist_dev, out_count_dev, nullptr,
                           (hipStream_t)stream);
}

int32_t vers_ivf_search_partial_dev(vers_ivf_t* h, const float* queries_dev, uint64_t ldq_floats, uint32_t b, uint32_t top_k,
                                    uint32_t nprobe, uint64_t* out_keys_dev, uint64_t* out_ids_dev, void* stream) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  if (top_k == 0) return fail(VERS_ERR_INVALID, "vers_ivf_search_partial_dev: top_k must be at least 1");
  if (b && (!queries_dev || ldq_floats < h->d || !out_keys_dev || !out_ids_dev))
    return fail(VERS_ERR_INVALID, "vers_ivf_search_partial_dev: bad arguments");
  std::shared_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  WsLease lease(h, true, (hipStream_t)stream);
  if (lease.rc) return lease.rc;
  if (int32_t rc = lease.order_on((hipStream_t)stream)) return rc;
  // distances and counts of the local part are scratch here: the cross-GPU merge recomputes them
  if (int32_t rc = ensure_out(h, (size_t)b * top_k, b)) return rc;
  return search_dev_locked(h, queries_dev, ldq_floats, b, top_k, nprobe, out_ids_dev, W->o_dist.as<float>(), W->o_cnt.as<uint32_t>(),
                           out_keys_dev, (hipStream_t)stream);
}

// partial search / partial exhaustive scan -> the one exchange -> merge, all on `stream` (vers_hip.h: vers_gather_t)
static int32_t sharded_common(vers_ivf_t* h, const vers_gather_t* g, const float* queries_dev, uint64_t ldq_floats, uint32_t b, uint32_t top_k,
                              uint32_t nprobe, int exhaustive_metric, uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_count_dev, void* stream) {
  const char* who = exhaustive_metric >= 0 ? "vers_ivf_search_exhaustive_sharded_dev" : "vers_ivf_search_sharded_dev";
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  if (top_k == 0) return fail(VERS_ERR_INVALID, std::string(who) + ": top_k must be at least 1");
  if (b && (!queries_dev || ldq_floats < h->d || !out_ids_dev || !out_dist_dev || !out_count_dev))
    return fail(VERS_ERR_INVALID, std::string(who) + ": bad arguments");
  if (exhaustive_metric > (int)VERS_METRIC_COSDIST) return fail(VERS_ERR_INVALID, "unknown metric");
  const uint32_t world = g ? g->world : 1u;
  if (g && (g->world == 0 || g->rank >= g->world || (g->world > 1 && !g->all_gather_async))) return fail(VERS_ERR_INVALID, std::string(who) + ": incomplete vers_gather_t");
  if (b == 0) return VERS_OK;
  std::shared_lock<std::shared_mutex> lk(h->index);
  if (world != h->world || (g && world > 1 && g->rank != h->rank))
    return fail(VERS_ERR_INVALID, std::string(who) + ": the handle is sharded as rank " + std::to_string(h->rank) + " of " + std::to_string(h->world) + ", the exchange says otherwise");
  DeviceGuard gd(h->device);
  hipStream_t st = (hipStream_t)stream;
  WsLease lease(h, true, st);
  if (lease.rc) return lease.rc;
  if (int32_t rc = lease.order_on(st)) return rc;
  const size_t part = (size_t)b * top_k;  // keys | ids of one rank
  // The exchange buffers FIRST.  Up to here a local failure (no workspace, no exchange buffer) leaves this rank unable to take part
  // in the batch's all-gather at all: the peers have queued theirs and wait -- the host must abort the communicator
  // (vers_rccl_abort) when a rank returns from here, there is nothing the library can send.
  const bool exchange = g != nullptr && g->all_gather_async != nullptr;
  {
    int32_t rc = W->g_send.reserve(2 * part * sizeof(uint64_t));
    if (!rc) rc = W->g_recv.reserve((size_t)world * 2 * part * sizeof(uint64_t));
    if (rc) {
      if (exchange && world > 1)
        fail(rc, std::string(who) + ": the exchange buffers could not be reserved -- this rank cannot join the batch's all-gather; abort the communicator (" + vers_last_error() + ")");
      return rc;
    }
  }
  // (no exchange -- g == nullptr, a single process: the partial is the gathered buffer.  A one-rank communicator still gathers:
  // the call is the same code path as with peers, which is how the RCCL leg is tested on a one-GPU box)
  uint64_t* mine = exchange ? W->g_send.as<uint64_t>() : W->g_recv.as<uint64_t>();
  // From here on a LOCAL failure (scratch reservation, a failed launch) must not skip the batch's ONLY collective: the peers have
  // already queued their ncclAllGather and would block forever, and every later collective on the communicator would be
  // mismatched.  The rank answers with a POISONED partial instead -- every key kKeyMax (it contributes nothing), first id word
  // kPoisonId -- so that the gather completes everywhere; every rank's merge sees the mark and latches kStPeerFailed in its
  // stream's status word (vers_ivf_poll -> VERS_ERR_COMM); this rank also returns its own error from the call.
  int32_t local_rc = VERS_OK;
  if (W->g_poisoned) {  // (this workspace's send buffer carried the poison mark once: a partial whose first query finds nothing leaves the word as it is)
    (void)hipMemsetAsync(mine + part, 0, sizeof(uint64_t), st);
    W->g_poisoned = false;
  }
  if (test_fail_sharded_ref().load() > 0 && test_fail_sharded_ref().fetch_sub(1) > 0) local_rc = fail(VERS_ERR_HIP, std::string(who) + ": injected local failure (test hook)");
  if (!local_rc) local_rc = [&]() -> int32_t {
    if (int32_t rc = ensure_out(h, part, b)) return rc;
    if (exhaustive_metric >= 0) {
      if (int32_t rc = exhaustive_dev_locked(h, queries_dev, ldq_floats, b, top_k, (uint32_t)exhaustive_metric, W->o_ids.as<uint64_t>(), W->o_dist.as<float>(),
                                             W->o_cnt.as<uint32_t>(), st)) return rc;
      hipLaunchKernelGGL(pack_exhaustive_keys_kernel, dim3((unsigned)((part + 255) / 256)), dim3(256), 0, st, (const uint64_t*)W->o_ids.as<uint64_t>(),
                         (const float*)W->o_dist.as<float>(), (const uint32_t*)W->o_cnt.as<uint32_t>(), b, top_k, mine, mine + part);
      VERS_HIP_TRY(hipGetLastError());
      return VERS_OK;
    }
    return search_dev_locked(h, queries_dev, ldq_floats, b, top_k, nprobe, mine + part, W->o_dist.as<float>(), W->o_cnt.as<uint32_t>(), mine, st);
  }();
  if (local_rc && !exchange) return local_rc;
  if (local_rc) {  // (stream order: behind whatever the failed search had already queued into `mine`)
    const std::string why = vers_last_error();
    const uint64_t mark = kPoisonId;
    (void)hipMemsetAsync(mine, 0xFF, 2 * part * sizeof(uint64_t), st);
    (void)hipMemcpyAsync(mine + part, &mark, sizeof(mark), hipMemcpyHostToDevice, st);  // (pageable source: copied before the call returns)
    W->g_poisoned = true;
    (void)hipGetLastError();
    fail(local_rc, why + " [this rank joined the batch's all-gather with a poisoned partial]");
  }
  if (exchange) {
    if (int32_t rc = g->all_gather_async(g->ctx, W->g_send.p, W->g_recv.p, 2 * part * sizeof(uint64_t), stream))
      return local_rc ? local_rc : fail(VERS_ERR_COMM, "vers_gather_t::all_gather_async reported failure (status " + std::to_string(rc) + ")");
  }
  hipLaunchKernelGGL(rank_merge_kernel, dim3(b), dim3(kWave), 0, st, (const uint64_t*)W->g_recv.as<uint64_t>(), (const uint64_t*)W->g_recv.as<uint64_t>() + part,
                     (uint64_t)(2 * part), world, b, top_k, (exhaustive_metric < 0 && nprobe == 0) ? 1 : 0, out_ids_dev, out_dist_dev, out_count_dev,
                     exchange ? W->st_word() : (uint32_t*)nullptr);
  if (local_rc) { (void)hipGetLastError(); return local_rc; }
  VERS_HIP_TRY(hipGetLastError());
  return VERS_OK;
}

int32_t vers_ivf_search_sharded_dev(vers_ivf_t* h, const vers_gather_t* g, const float* queries_dev, uint64_t ldq_floats, uint32_t b, uint32_t top_k,
                                    uint32_t nprobe, uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_count_dev, void* stream) {
  return sharded_common(h, g, queries_dev, ldq_floats, b, top_k, nprobe, -1, out_ids_dev, out_dist_dev, out_count_dev, stream);
}

int32_t vers_ivf_search_exhaustive_sharded_dev(vers_ivf_t* h, const vers_gather_t* g, const float* queries_dev, uint64_t ldq_floats, uint32_t b,
                                               uint32_t top_k, uint32_t metric, uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_count_dev,
                                               void* stream) {
  if (metric > VERS_METRIC_COSDIST) return fail(VERS_ERR_INVALID, "unknown metric");
  return sharded_common(h, g, queries_dev, ldq_floats, b, top_k, 1, (int)metric, out_ids_dev, out_dist_dev, out_count_dev, stream);
}

int32_t vers_ivf_coarse_ahead_dev(vers_ivf_t* h, const float* queries_dev, uint64_t ldq_floats, uint32_t b, uint32_t nprobe, void* stream) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  if (b && (!queries_dev || ldq_floats < h->d)) return fail(VERS_ERR_INVALID, "vers_ivf_coarse_ahead_dev: bad arguments");
  std::lock_guard<std::mutex> lk(h->pool_mu);
  (void)stream;  // (the look-ahead is ordered behind the list scan of the NEXT search on this handle, on that search's stream)
  h->pending.set = b != 0 && nprobe != 0;
  h->pending.q_dev = queries_dev; h->pending.ldq_in = ldq_floats; h->pending.b = b; h->pending.nprobe = nprobe;
  return VERS_OK;
}

int32_t vers_topk_merge_dev(const uint64_t* keys_dev, const uint64_t* ids_dev, uint64_t rank_stride, uint32_t world, uint32_t b,
                            uint32_t top_k, uint32_t nprobe, uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_count_dev,
                            void* stream) {
  if (rank_stride < (uint64_t)b * top_k) return fail(VERS_ERR_INVALID, "vers_topk_merge_dev: rank_stride < b * top_k");
  if (world == 0 || top_k == 0 || (b && (!keys_dev || !ids_dev || !out_ids_dev || !out_dist_dev || !out_count_dev)))
    return fail(VERS_ERR_INVALID, "vers_topk_merge_dev: bad arguments");
  if (b == 0) return VERS_OK;
  hipLaunchKernelGGL(rank_merge_kernel, dim3(b), dim3(kWave), 0, (hipStream_t)stream, keys_dev, ids_dev, rank_stride, world, b, top_k,
                     nprobe == 0 ? 1 : 0, out_ids_dev, out_dist_dev, out_count_dev);
  VERS_HIP_TRY(hipGetLastError());
  return VERS_OK;
}

int32_t vers_ivf_search(vers_ivf_t* h, const float* queries, uint64_t q_stride_bytes, uint32_t b, uint32_t top_k,
                        uint32_t nprobe, uint64_t* out_ids, float* out_dist, uint32_t* out_count) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  if (b && (!queries || q_stride_bytes < (uint64_t)h->d * 4 || q_stride_bytes % 4 || !out_count || (top_k && (!out_ids || !out_dist))))
    return fail(VERS_ERR_INVALID, "vers_ivf_search: bad arguments");
  if (b == 0) return VERS_OK;
  std::shared_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  WsLease lease(h);
  if (lease.rc) return lease.rc;
  HostStatusSlot slot(h);
  HostIo io;
  if (int32_t rc = host_io_begin(h, queries, q_stride_bytes, b, top_k, io, b == 1 && top_k > 0)) return rc;
  lease.st = W->io_stream;
  // Reference mode ranks only as many lists as the spill may need: 16 first (the merge of the coarse partial lists
  // and the plan are what a single-query call waits for), then 48, then 64 with the exact coarse quantiser, and as
  // the last resort ALL of them (ivfflat.rs:166-195 walks the ranked lists as far as it must -- many empty lists with
  // zero centroids make that real); the batch then goes in slices so that the (query, list) tables stay small.
  int32_t rc = VERS_OK;
  for (int attempt = nprobe == 0 ? 0 : 1; attempt < 4; ++attempt) {
    W->ref_shallow = attempt == 0;
    W->ref_deep = attempt == 2;
    W->ref_all = attempt == 3;
    // every attempt starts from zeros: entries past a query's count must not carry a previous attempt's values
    if (io.direct) {  // (no kernel of this workspace that WRITES is in flight: the previous call ended with its published status or a synchronisation;
                      // a fallback_kernel launch of that call may still be queued -- it returns before its first store when nothing was queued for it:
                      // the publisher invariant in finish.hip.h)
      std::memset((char*)io.ids_dev - io.ids_off, 0, io.out_bytes);
      W->st_host = (uint32_t*)((char*)io.ids_dev - io.ids_off + io.st_off);
      *reinterpret_cast<volatile uint32_t*>(W->st_host) = kStNotYet;  // (the last merge launch overwrites it with the status: host_io_end spins on it)
    } else {
      VERS_HIP_TRY(hipMemsetAsync(W->io_out.p, 0, io.out_bytes, W->io_stream));
    }
    const uint32_t slice = attempt == 3 ? std::max<uint32_t>(1u, 65536u / std::max<uint32_t>(1u, h->k)) : b;
    rc = VERS_OK;
    for (uint32_t q0 = 0; q0 < b && rc == VERS_OK; q0 += slice) {
      const uint32_t bq = std::min(slice, b - q0);
      rc = search_dev_locked(h, io.q_dev + (size_t)q0 * h->d, h->d, bq, top_k, nprobe, io.ids_dev + (size_t)q0 * top_k,
                             io.dist_dev + (size_t)q0 * top_k, io.cnt_dev + q0, nullptr, W->io_stream);
    }
    W->ref_shallow = W->ref_deep = W->ref_all = false;
    W->st_host = nullptr;
    if (rc) return rc;
    rc = host_io_end(h, io, b, top_k, out_ids, out_dist, out_count);
    if (rc != kRetrySpill) break;
    if ((attempt == 0 && h->k <= 16) || (attempt == 1 && h->k <= 48) || (attempt == 2 && h->k <= 64)) break;  // every list was ranked already
  }
  return rc == kRetrySpill ? VERS_ERR_INVALID : rc;
}

int32_t vers_ivf_search_exhaustive_dev(vers_ivf_t* h, const float* queries_dev, uint64_t ldq_floats, uint32_t b, uint32_t top_k,
                                       uint32_t metric, uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_count_dev,
                                       void* stream) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  if (metric > VERS_METRIC_COSDIST) return fail(VERS_ERR_INVALID, "unknown metric");
  if (b && (!queries_dev || ldq_floats < h->d || !out_count_dev || (top_k && (!out_ids_dev || !out_dist_dev))))
    return fail(VERS_ERR_INVALID, "vers_ivf_search_exhaustive_dev: bad arguments");
  std::shared_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  WsLease lease(h, true, (hipStream_t)stream);
  if (lease.rc) return lease.rc;
  if (int32_t rc = lease.order_on((hipStream_t)stream)) return rc;
  return exhaustive_dev_locked(h, queries_dev, ldq_floats, b, top_k, metric, out_ids_dev, out_dist_dev, out_count_dev,
                               (hipStream_t)stream);
}

int32_t vers_ivf_search_exhaustive_partial_dev(vers_ivf_t* h, const float* queries_dev, uint64_t ldq_floats, uint32_t b, uint32_t top_k,
                                               uint32_t metric, uint64_t* out_keys_dev, uint64_t* out_ids_dev, void* stream) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  if (top_k == 0 || metric > VERS_METRIC_COSDIST) return fail(VERS_ERR_INVALID, "unsupported top_k / metric");
  if (b && (!queries_dev || ldq_floats < h->d || !out_keys_dev || !out_ids_dev))
    return fail(VERS_ERR_INVALID, "vers_ivf_search_exhaustive_partial_dev: bad arguments");
  if (b == 0) return VERS_OK;
  std::shared_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  WsLease lease(h, true, (hipStream_t)stream);
  if (lease.rc) return lease.rc;
  if (int32_t rc = lease.order_on((hipStream_t)stream)) return rc;
  if (int32_t rc = ensure_out(h, (size_t)b * top_k, b)) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (int32_t rc = exhaustive_dev_locked(h, queries_dev, ldq_floats, b, top_k, metric, W->o_ids.as<uint64_t>(), W->o_dist.as<float>(),
                                         W->o_cnt.as<uint32_t>(), st)) return rc;
  hipLaunchKernelGGL(pack_exhaustive_keys_kernel, dim3((b * top_k + 255) / 256), dim3(256), 0, st, (const uint64_t*)W->o_ids.as<uint64_t>(),
                     (const float*)W->o_dist.as<float>(), (const uint32_t*)W->o_cnt.as<uint32_t>(), b, top_k, out_keys_dev, out_ids_dev);
  VERS_HIP_TRY(hipGetLastError());
  return VERS_OK;
}

int32_t vers_ivf_search_exhaustive(vers_ivf_t* h, const float* queries, uint64_t q_stride_bytes, uint32_t b, uint32_t top_k,
                                   uint32_t metric, uint64_t* out_ids, float* out_dist, uint32_t* out_count) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  if (metric > VERS_METRIC_COSDIST) return fail(VERS_ERR_INVALID, "unknown metric");
  if (b && (!queries || q_stride_bytes < (uint64_t)h->d * 4 || q_stride_bytes % 4 || !out_count || (top_k && (!out_ids || !out_dist))))
    return fail(VERS_ERR_INVALID, "vers_ivf_search_exhaustive: bad arguments");
  if (b == 0) return VERS_OK;
  std::shared_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  WsLease lease(h);
  if (lease.rc) return lease.rc;
  HostStatusSlot slot(h);
  HostIo io;
  if (int32_t rc = host_io_begin(h, queries, q_stride_bytes, b, top_k, io)) return rc;
  lease.st = W->io_stream;
  VERS_HIP_TRY(hipMemsetAsync(W->io_out.p, 0, io.out_bytes, W->io_stream));  // entries past a query's count come back as zeros
  if (int32_t rc = exhaustive_dev_locked(h, io.q_dev, h->d, b, top_k, metric, io.ids_dev, io.dist_dev, io.cnt_dev, W->io_stream)) return rc;
  const int32_t rc = host_io_end(h, io, b, top_k, out_ids, out_dist, out_count);
  return rc == kRetrySpill ? VERS_ERR_INVALID : rc;
}

}  // extern "C"
