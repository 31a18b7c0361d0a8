// ivf_hooks.hip -- measurement hooks of the IVFFlat handle: timings of the last launches, planner / certificate statistics.
// (The TEST hooks -- partial-list dump, slack poisoning, lane networks, MFMA probe, RCCL stand-in -- are NOT part of this library:
// csrc/testhooks/, include/vers_hip_test.h, libvers_hip_test.so.)
#include "ivf_handle.hpp"
#include "prescan.hip.h"


extern "C" {

int32_t vers_ivf_last_scan(vers_ivf_t* h, float* out_ms, uint64_t* out_union_rows, uint64_t* out_streamed_rows,
                           uint32_t* out_items) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  UseLastWs use_ws(h);
  if (!use_ws.ok) return fail(VERS_ERR_INVALID, "no search has run on this handle");
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  if (W->ev_count == 0 || !W->tot_valid) return fail(VERS_ERR_INVALID, "no list scan has been launched on this handle");
  DeviceGuard g(h->device);
  const uint32_t slot = (uint32_t)((W->ev_count - 1) % SearchWs::kEvRing);
  VERS_HIP_TRY(hipEventSynchronize(W->ev1[slot]));
  if (out_ms) VERS_HIP_TRY(hipEventElapsedTime(out_ms, W->ev0[slot], W->ev1[slot]));
  const GroupTotals* tot = W->tot_dev;
  if (!tot) return fail(VERS_ERR_INVALID, "vers_ivf_last_scan: no list scan has run on this handle");
  GroupTotals t;
  VERS_HIP_TRY(hipMemcpy(&t, tot, sizeof(t), hipMemcpyDeviceToHost));
  if (out_union_rows) *out_union_rows = t.union_rows;
  if (out_streamed_rows) *out_streamed_rows = t.streamed_rows;
  if (out_items) *out_items = t.n_items;
  if ((scan_debug_flags() & 16u) && W->stamps.p) {  // diagnosis only: per-wave phase cycles of the last launch
    unsigned long long sv[64] = {};
    VERS_HIP_TRY(hipMemcpy(sv, W->stamps.p, std::min<size_t>(512, W->stamps.cap), hipMemcpyDeviceToHost));
    if (W->stamps.cap >= 512 && sv[37])
      fprintf(stderr, "[vers stamps] single-query coarse + plan kernel, its last block %llu (us): loads + products %.2f  chains %.2f  sort + publish %.2f  "
              "acquire %.2f  merge + plan %.2f\n", sv[38], (sv[33] - sv[32]) / 100.0, (sv[34] - sv[33]) / 100.0, (sv[35] - sv[34]) / 100.0,
              (sv[36] - sv[35]) / 100.0, (sv[37] - sv[36]) / 100.0);
    if (sv[50]) fprintf(stderr, "[vers stamps]   merge %.2f  list tables + scan %.2f  plan stores %.2f  rest %.2f\n", (sv[48] - sv[36]) / 100.0, (sv[49] - sv[48]) / 100.0, (sv[50] - sv[49]) / 100.0, (sv[37] - sv[50]) / 100.0);
    if (sv[27])
      fprintf(stderr, "[vers stamps] coarse select, per query avg cycles: select %.0f  exact re-score %.0f  sort+certify+emit %.0f\n",
              (double)sv[24] / sv[27], (double)sv[25] / sv[27], (double)sv[26] / sv[27]);
    if (sv[27])
      fprintf(stderr, "[vers stamps]   of the selection: loads + order bits %.0f  bracket %.0f  compact + sort %.0f\n", (double)sv[28] / sv[27], (double)sv[29] / sv[27],
              (double)sv[30] / sv[27]);
    if (sv[56])
      fprintf(stderr, "[vers stamps] exact finish, per query avg cycles: merge of the partial lists %.0f  certificate + storage rows %.0f  gather + chains %.0f  "
              "sort + emit %.0f; survivors per query %.1f\n", (double)sv[52] / sv[56], (double)sv[53] / sv[56], (double)sv[54] / sv[56], (double)sv[55] / sv[56],
              (double)sv[57] / sv[56]);
    if (sv[56] && sv[59])
      fprintf(stderr, "[vers stamps]   single query's merge by counting: preamble %.0f  key + minima loads %.0f  T (sort + folds) %.0f  filter %.0f  ranks %.0f\n", (double)sv[58] / sv[56],
              (double)sv[59] / sv[56], (double)sv[60] / sv[56], (double)sv[61] / sv[56], (double)sv[62] / sv[56]);
    if (sv[19])
      fprintf(stderr, "[vers stamps] group / scatter kernel, block 0 (us): fill + prefix sums %.1f  scatter %.1f  items %.1f\n",
              (sv[17] - sv[16]) / 100.0, (sv[18] - sv[17]) / 100.0, (sv[19] - sv[18]) / 100.0);
    if (sv[10])
      fprintf(stderr, "[vers stamps] matrix-core scan, per item avg cycles: prologue %.0f  step loop %.0f (of which issuing loads %.0f)  epilogue %.0f\n",
              (double)sv[9] / sv[4], (double)sv[10] / sv[4], (double)sv[8] / sv[4], (double)sv[11] / sv[4]);
    fprintf(stderr, "[vers stamps] items %llu: per item avg cycles: wait-for-loads %.0f  math %.0f  fold %.0f | per wave-quad-slot (%llu): stage %.0f  barrier-wait %.0f\n",
            sv[4], sv[4] ? (double)sv[0] / sv[4] : 0.0, sv[4] ? (double)sv[1] / sv[4] : 0.0, sv[4] ? (double)sv[2] / sv[4] : 0.0,
            sv[6], sv[6] ? (double)sv[3] / sv[6] : 0.0, sv[6] ? (double)sv[5] / sv[6] : 0.0);
    fprintf(stderr, "[vers stamps] list merges under a lock %llu, candidates offered to them %llu\n", sv[12], sv[13]);
    fprintf(stderr, "[vers stamps] shader clock during the kernel: %.0f MHz\n", (double)sv[7] / (double)(1 << 20) * 100.0);
  }
  return VERS_OK;
}

int32_t vers_ivf_last_coarse_ms(vers_ivf_t* h, float* out_gemm_ms, float* out_select_ms) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  UseLastWs use_ws(h);
  if (!use_ws.ok) return fail(VERS_ERR_INVALID, "no search has run on this handle");
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  if (!W->evc_valid) return fail(VERS_ERR_INVALID, "no batched coarse quantiser has run on the matrix cores on this handle");
  DeviceGuard g(h->device);
  VERS_HIP_TRY(hipEventSynchronize(W->evc[2]));
  if (out_gemm_ms) VERS_HIP_TRY(hipEventElapsedTime(out_gemm_ms, W->evc[0], W->evc[1]));
  if (out_select_ms) VERS_HIP_TRY(hipEventElapsedTime(out_select_ms, W->evc[1], W->evc[2]));
  return VERS_OK;
}

int32_t vers_ivf_last_finish_ms(vers_ivf_t* h, float* out_ms) {
  if (!h || !out_ms) return fail(VERS_ERR_INVALID, "bad arguments");
  UseLastWs use_ws(h);
  if (!use_ws.ok || !W->evf_valid) return fail(VERS_ERR_INVALID, "no matrix-core batch with event records has run on this handle");
  DeviceGuard g(h->device);
  VERS_HIP_TRY(hipEventSynchronize(W->evf));
  VERS_HIP_TRY(hipEventElapsedTime(out_ms, W->ev1[W->evf_slot], W->evf));
  return VERS_OK;
}

int32_t vers_ivf_coarse_stats(vers_ivf_t* h, uint64_t* out_mfma_batches, uint64_t* out_fallback_queries) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  DeviceGuard g(h->device);
  uint32_t fb = 0;
  if (h->coarse_stat.p) VERS_HIP_TRY(hipMemcpy(&fb, h->coarse_stat.p, 4, hipMemcpyDeviceToHost));
  if (out_mfma_batches) *out_mfma_batches = h->mfma_batches;
  if (out_fallback_queries) *out_fallback_queries = fb;
  return VERS_OK;
}

int32_t vers_ivf_prescan_stats(vers_ivf_t* h, uint64_t* out_batches, uint64_t* out_fallback_queries) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  DeviceGuard g(h->device);
  uint32_t fb = 0;
  if (h->pre_misc.p) VERS_HIP_TRY(hipMemcpy(&fb, h->pre_misc.as<uint32_t>() + 1, 4, hipMemcpyDeviceToHost));
  if (out_batches) *out_batches = h->pre_batches;
  if (out_fallback_queries) *out_fallback_queries = fb;
  return VERS_OK;
}

int32_t vers_ivf_scan_times(vers_ivf_t* h, float* out_ms, uint32_t cap, uint32_t* out_n, int32_t reset) {
  if (!h || !out_n || (cap && !out_ms)) return fail(VERS_ERR_INVALID, "bad arguments");
  DeviceGuard g(h->device);
  // every workspace's ring (batches kept in flight on several streams lease one each); within a ring oldest first.  The
  // caller has stopped issuing searches (a measurement hook): the rings are read without leasing.
  std::vector<SearchWs*> all;
  {
    std::lock_guard<std::mutex> lk(h->pool_mu);
    for (auto& w : h->pool) all.push_back(w.get());
  }
  uint32_t n = 0;
  for (SearchWs* w : all) {
    if (!w->done) continue;  // never initialised
    const uint64_t have = std::min<uint64_t>(w->ev_count, SearchWs::kEvRing);
    for (uint64_t i = 0; i < have && n < cap; ++i) {
      const uint32_t slot = (uint32_t)((w->ev_count - have + i) % SearchWs::kEvRing);
      VERS_HIP_TRY(hipEventSynchronize(w->ev1[slot]));
      VERS_HIP_TRY(hipEventElapsedTime(&out_ms[n], w->ev0[slot], w->ev1[slot]));
      ++n;
    }
    if (reset) w->ev_count = 0;
  }
  *out_n = n;
  return VERS_OK;
}

}  // extern "C"
