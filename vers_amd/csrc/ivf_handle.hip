// ivf_handle.hip -- lifecycle of the IVFFlat handle, its workspace pool and status words, options and read-only getters.
// (ivf_handle.hpp has the map of the index's translation units.)
#include "ivf_handle.hpp"

namespace vers {

// Longest-processing-time assignment of whole inverted lists to GPUs: lists by (length desc, index asc),
// each to the currently least loaded rank (ties -> lowest rank).  Deterministic, so every process
// derives the same plan from the same list lengths without talking to the others.
void shard_plan(const uint64_t* lens, uint64_t k, uint32_t world, uint8_t* owner) {
  std::vector<uint64_t> order(k);
  for (uint64_t i = 0; i < k; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return lens[a] > lens[b]; });
  std::vector<uint64_t> load(world ? world : 1, 0);
  for (uint64_t i : order) {
    uint32_t best = 0;
    for (uint32_t r = 1; r < world; ++r)
      if (load[r] < load[best]) best = r;
    owner[i] = (uint8_t)best;
    load[best] += lens[i];
  }
}

}  // namespace vers

thread_local SearchWs* W = nullptr;

namespace vers {
namespace ivf {

// ---- workspaces ------------------------------------------------------------------------------------------------
int32_t ws_init(SearchWs& w) {
  if (int32_t rc = w.status.reserve(16)) return rc;
  VERS_HIP_TRY(hipMemset(w.status.p, 0, 16));
  if (scan_debug_flags() & 16u) {  // diagnosis: in-kernel phase stamps
    if (int32_t rc = w.stamps.reserve(512)) return rc;
    VERS_HIP_TRY(hipMemset(w.stamps.p, 0, 512));
  }
  for (uint32_t i = 0; i < SearchWs::kEvRing; ++i) {
    VERS_HIP_TRY(hipEventCreate(&w.ev0[i]));
    VERS_HIP_TRY(hipEventCreate(&w.ev1[i]));
  }
  for (auto& e : w.evc) VERS_HIP_TRY(hipEventCreate(&e));
  VERS_HIP_TRY(hipEventCreate(&w.evf));
  VERS_HIP_TRY(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
  return VERS_OK;
}
void ws_destroy(SearchWs& w) {
  for (uint32_t i = 0; i < SearchWs::kEvRing; ++i) {
    if (w.ev0[i]) (void)hipEventDestroy(w.ev0[i]);
    if (w.ev1[i]) (void)hipEventDestroy(w.ev1[i]);
  }
  for (auto& e : w.evc)
    if (e) (void)hipEventDestroy(e);
  if (w.evf) (void)hipEventDestroy(w.evf);
  if (w.done) (void)hipEventDestroy(w.done);
  if (w.io_pin) (void)hipHostFree(w.io_pin);
  if (w.io_stream) (void)hipStreamDestroy(w.io_stream);
  if (w.ahead_stream) {
    (void)hipStreamSynchronize(w.ahead_stream);
    (void)hipStreamDestroy(w.ahead_stream);
    (void)hipEventDestroy(w.ahead_in);
    for (auto& a : w.ahead) { (void)hipEventDestroy(a.ready); (void)hipEventDestroy(a.freed); }
  }
}

int32_t sync_status(vers_ivf* h, hipStream_t st) {  // the word of the _dev calls queued on `st` (vers_ivf::stream_word), read and cleared on `st`
  uint32_t *word = nullptr, *pin = nullptr;
  std::mutex* poll_mu = nullptr;
  if (int32_t rc = h->stream_word(st, &word, &pin, &poll_mu)) return rc;
  uint32_t s = 0;
  {
    // (the stream's own landing word, under the word's own lock: searches and polls on other streams go on while this waits)
    std::lock_guard<std::mutex> lk(*poll_mu);
    VERS_HIP_TRY(hipMemcpyAsync(pin, word, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    VERS_HIP_TRY(hipMemsetAsync(word, 0, sizeof(uint32_t), st));  // stream order: behind every kernel that could set it, ahead of the next call's
    VERS_HIP_TRY(hipStreamSynchronize(st));
    s = *reinterpret_cast<volatile uint32_t*>(pin);
  }
  if (!s) return VERS_OK;
  if (s & kStPeerFailed)
    return fail(VERS_ERR_COMM, "a rank of the sharded search failed locally and sent a poisoned partial: results of batches on this stream miss that rank's lists");
  if (s & kStNaN) return fail(VERS_ERR_NAN, "NaN distance (the reference panics in partial_cmp().unwrap())");
  if (s & kStInsufficient)
    return fail(VERS_ERR_INSUFFICIENT, "fewer than top_k vectors reachable (reference: index out of bounds, ivfflat.rs:169)");
  if (s & kStSpillTooDeep) {
    fail(VERS_ERR_INVALID, "search_approximate spills past the lists this device-pointer call ranked (48): the host-pointer entry point retries with every list");
    return kRetrySpill;
  }
  return VERS_OK;
}

// maps (and clears) the device status word of a finished search: the reference's panics
int32_t status_to_rc(vers_ivf* h, uint32_t s, uint32_t slot) {
  if (s) {
    VERS_HIP_TRY(hipMemset(W->status.as<uint32_t>() + slot, 0, sizeof(s)));
    if (s & kStNaN) return fail(VERS_ERR_NAN, "NaN distance (the reference panics in partial_cmp().unwrap())");
    if (s & kStInsufficient)
      return fail(VERS_ERR_INSUFFICIENT, "fewer than top_k vectors reachable (reference: index out of bounds, ivfflat.rs:169)");
    if (s & kStSpillTooDeep) {
      fail(VERS_ERR_INVALID, "search_approximate spills past the lists this device-pointer call ranked (48): the host-pointer entry point retries with every list");
      return kRetrySpill;
    }
  }
  return VERS_OK;
}

}  // namespace ivf
}  // namespace vers

extern "C" {

int32_t vers_ivf_create(int32_t device, uint32_t d, vers_ivf_t** out) {
  if (!out || d == 0) return fail(VERS_ERR_INVALID, "vers_ivf_create: bad arguments");
  int cnt = 0;
  VERS_HIP_TRY(hipGetDeviceCount(&cnt));
  if (device < 0 || device >= cnt) return fail(VERS_ERR_INVALID, "vers_ivf_create: no such device");
  DeviceGuard g(device);
  vers_ivf* h = new (std::nothrow) vers_ivf();
  if (!h) return fail(VERS_ERR_INVALID, "out of host memory");
  h->device = device;
  h->d = d;
  h->ldx = round_up(d, 4);
  h->ld = round_up(d, kColAlign);
  h->ldq = h->ld;
  const int32_t rc = [&]() -> int32_t {
    hipDeviceProp_t prop;
    VERS_HIP_TRY(hipGetDeviceProperties(&prop, device));
    h->n_cu = prop.multiProcessorCount;
    return VERS_OK;  // (workspaces -- status words, events, scratch -- are made when calls lease them)
  }();
  if (rc != VERS_OK) {  // nothing half-made leaks: destroy releases whatever was created
    (void)vers_ivf_destroy(h);
    return rc;
  }
  *out = h;
  return VERS_OK;
}

int32_t vers_ivf_set_metric(vers_ivf_t* h, uint32_t metric) {
  if (!h || metric > VERS_METRIC_COSDIST) return fail(VERS_ERR_INVALID, "vers_ivf_set_metric: bad arguments");
  std::unique_lock<std::shared_mutex> lk(h->index);
  if (h->k != 0 && (int)metric != h->metric) return fail(VERS_ERR_INVALID, "vers_ivf_set_metric: call before build / upload");
  h->metric = (int)metric;
  return VERS_OK;
}

int32_t vers_ivf_get_metric(vers_ivf_t* h, uint32_t* out_metric) {
  if (!h || !out_metric) return fail(VERS_ERR_INVALID, "bad arguments");
  *out_metric = (uint32_t)h->metric;
  return VERS_OK;
}

int32_t vers_ivf_destroy(vers_ivf_t* h) {
  if (!h) return VERS_OK;
  DeviceGuard g(h->device);
  (void)hipDeviceSynchronize();
  for (auto& w : h->pool) ws_destroy(*w);
  if (h->fail_watch) (void)hipHostFree(h->fail_watch);
  if (h->st_pin) (void)hipHostFree(h->st_pin);
  h->up.close();
  delete h;
  return VERS_OK;
}

int32_t vers_set_option(const char* name, int64_t value) {
  if (!name) return fail(VERS_ERR_INVALID, "vers_set_option: null name");
  if (std::strcmp(name, "gemm_x3") == 0) { set_gemm_x3_mask((int)value); return VERS_OK; }
  if (std::strcmp(name, "shadow") == 0) { shadow_mode_ref().store(value != 0 ? 1 : 0); return VERS_OK; }
  if (std::strcmp(name, "pre_min_batch") == 0) { pre_min_batch_ref().store(value < 2 ? 2u : (uint32_t)std::min<int64_t>(value, 0x7FFFFFFF)); return VERS_OK; }
  if (std::strcmp(name, "test_fail_sharded") == 0) { test_fail_sharded_ref().store((int)value); return VERS_OK; }  // TEST HOOK: the next `value` sharded searches of this process fail locally
  if (std::strcmp(name, "scan_reserve_cus") == 0) { scan_reserve_cus_ref().store((int)std::max<int64_t>(-1, std::min<int64_t>(value, 1 << 20))); return VERS_OK; }
  if (std::strcmp(name, "single_shadow") == 0) { single_shadow_ref().store(value != 0 ? 1 : 0); return VERS_OK; }
  if (std::strcmp(name, "host_spin") == 0) { host_spin_ref().store(value != 0 ? 1 : 0); return VERS_OK; }
  if (std::strcmp(name, "scan_events") == 0) { scan_events_ref().store(value < 0 || value > 2 ? 2 : (int)value); return VERS_OK; }
  if (opt_set(name, value)) return VERS_OK;  // every other switch: read where it is used (core.hip's table)
  return fail(VERS_ERR_INVALID, std::string("vers_set_option: unknown option ") + name);
}

int32_t vers_mem_stats(uint64_t* out_bytes_now, uint64_t* out_bytes_peak, int32_t reset_peak) {
  dev_mem_stats(out_bytes_now, out_bytes_peak, reset_peak != 0);
  return VERS_OK;
}

int32_t vers_shard_plan(const uint64_t* list_lengths, uint64_t k, uint32_t world, uint8_t* out_owner) {
  if ((k && (!list_lengths || !out_owner)) || world == 0 || world > 255) return fail(VERS_ERR_INVALID, "vers_shard_plan: bad arguments");
  shard_plan(list_lengths, k, world, out_owner);
  return VERS_OK;
}

int32_t vers_ivf_set_shard(vers_ivf_t* h, uint32_t rank, uint32_t world) {
  if (!h || world == 0 || world > 255 || rank >= world) return fail(VERS_ERR_INVALID, "vers_ivf_set_shard: bad arguments");
  std::unique_lock<std::shared_mutex> lk(h->index);
  if (h->k != 0 || h->up.open) return fail(VERS_ERR_INVALID, "vers_ivf_set_shard: call before build / upload");
  h->rank = rank;
  h->world = world;
  return VERS_OK;
}

int32_t vers_ivf_owners(vers_ivf_t* h, uint8_t* out_owner) {
  if (!h || (h->k && !out_owner)) return fail(VERS_ERR_INVALID, "bad arguments");
  for (uint32_t c = 0; c < h->k; ++c) out_owner[c] = h->h_owner[c];
  return VERS_OK;
}

int32_t vers_ivf_poll(vers_ivf_t* h, void* stream) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  DeviceGuard g(h->device);
  const int32_t rc = sync_status(h, (hipStream_t)stream);
  return rc == kRetrySpill ? VERS_ERR_INVALID : rc;
}

int32_t vers_ivf_info(vers_ivf_t* h, uint64_t* out_n, uint64_t* out_k, uint64_t* out_max_list_len) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  if (out_n) *out_n = h->n_total;
  if (out_k) *out_k = h->k;
  if (out_max_list_len) *out_max_list_len = h->max_len;
  return VERS_OK;
}

int32_t vers_ivf_list_lengths(vers_ivf_t* h, uint64_t* out_lengths) {
  if (!h || (h->k && !out_lengths)) return fail(VERS_ERR_INVALID, "bad arguments");
  for (uint32_t c = 0; c < h->k; ++c) out_lengths[c] = h->h_len[c];
  return VERS_OK;
}

int32_t vers_ivf_shadow_state(vers_ivf_t* h, int32_t* out_active, uint64_t* out_bytes) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  if (out_active) *out_active = (shadow_mode() != 0 && h->shadow_valid && h->rows_bf.p != nullptr && !h->shadow_off) ? 1 : 0;
  if (out_bytes) *out_bytes = h->rows_bf.p ? (uint64_t)h->rows_bf.cap : 0;
  return VERS_OK;
}

int32_t vers_ivf_layout_bytes(vers_ivf_t* h, uint64_t* out_rows, uint64_t* out_shadow, uint64_t* out_rowmajor) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  std::shared_lock<std::shared_mutex> lk(h->index);
  if (out_rows) *out_rows = h->rows.p ? (uint64_t)h->cap_rows * h->ld * sizeof(float) : 0;
  if (out_shadow) *out_shadow = h->rows_bf.p && h->shadow_valid ? (uint64_t)h->cap_rows * h->ld * sizeof(uint16_t) : 0;
  if (out_rowmajor) *out_rowmajor = h->rows_rm.p ? (uint64_t)h->cap_rows * h->ld * sizeof(float) : 0;
  return VERS_OK;
}

int32_t vers_ivf_get_list(vers_ivf_t* h, uint64_t cluster, float* out_rows, uint64_t row_stride_bytes, uint64_t* out_ids,
                          uint64_t cap_rows, uint64_t* out_len) {
  if (!h || cluster >= h->k || !out_len) return fail(VERS_ERR_INVALID, "vers_ivf_get_list: bad arguments");
  std::shared_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  const uint32_t len = h->h_len[cluster];
  *out_len = len;
  if (!out_rows && !out_ids) return VERS_OK;
  if (h->h_owner[cluster] != h->rank) return fail(VERS_ERR_INVALID, "vers_ivf_get_list: list is stored on another GPU (see vers_ivf_owners)");
  if (cap_rows < len || (out_rows && row_stride_bytes < (uint64_t)h->d * 4)) return fail(VERS_ERR_INVALID, "vers_ivf_get_list: buffer too small");
  if (len == 0) return VERS_OK;
  if (out_rows) {
    DevBuf tmp;
    if (int32_t rc = tmp.reserve((size_t)len * h->d * sizeof(float))) return rc;
    if (int32_t rc = launch_from_blocked(h->rows.as<float>(), h->ld, h->h_off[cluster], len, h->d, tmp.as<float>(), h->d, nullptr))
      return rc;
    VERS_HIP_TRY(hipMemcpy2D(out_rows, row_stride_bytes, tmp.p, (size_t)h->d * 4, (size_t)h->d * 4, len, hipMemcpyDeviceToHost));
  }
  if (out_ids) {
    std::vector<uint32_t> ids(len);
    VERS_HIP_TRY(hipMemcpy(ids.data(), h->row_ids.as<uint32_t>() + h->h_off[cluster], (size_t)len * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < len; ++i) out_ids[i] = ids[i];
  }
  return VERS_OK;
}

int32_t vers_ivf_get_centroids(vers_ivf_t* h, float* out_centroids, uint64_t c_stride_bytes) {
  if (!h || (h->k && !out_centroids) || c_stride_bytes < (uint64_t)h->d * 4) return fail(VERS_ERR_INVALID, "bad arguments");
  if (h->k == 0) return VERS_OK;
  DeviceGuard g(h->device);
  VERS_HIP_TRY(hipMemcpy2D(out_centroids, c_stride_bytes, h->centroids.p, (size_t)h->ldx * 4, (size_t)h->d * 4, h->k,
                           hipMemcpyDeviceToHost));
  return VERS_OK;
}

}  // extern "C"
