// flat.hip -- brute-force scan behind vers_flat_* (replaces utils::search_exhaustive,
// /root/reference/vers/src/utils.rs:68-82).
#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

#include "flat_shadow.hpp"
#include "scan.hip.h"
#include "util.hip.h"

namespace vers {

// item = (row segment, query group).  slot(q, seg) = q * n_segs + seg.  For QG > 1 the segments of a query
// group are padded to a multiple of 4 (empty items) so that every quad of items shares one query block.
template <int QG, bool SEQ_IDS>
struct FlatSrc {
  static constexpr bool kSeqIds = SEQ_IDS;
  static constexpr bool kStreamOnce = true;
  const float* rows;  // blocked tiles
  uint64_t n;
  uint32_t ld;
  uint32_t seg_rows, n_segs, n_segs_pad;  // seg_rows is a multiple of 64
  const float* queries;  // QG == 1: [b][ldq]; else interleaved blocks [ceil(b/QG)][ldq][QG]; zero padded
  uint32_t ldq, b;
  uint64_t* partials;
  uint32_t k;
  const uint32_t* ids;  // SEQ_IDS: row -> seq (vec ids of a permuted corpus)

  __device__ __forceinline__ uint32_t n_items() const { return n_segs_pad * ((b + QG - 1) / QG); }
  __device__ __forceinline__ void get(uint32_t it, ItemView<QG>& v) const {
    const uint32_t seg = it % n_segs_pad, qg = it / n_segs_pad;
    const bool real = seg < n_segs;
    const uint64_t row0 = real ? (uint64_t)seg * seg_rows : 0;
    v.rows = rows + row0 * ld;
    v.nrows = real ? (uint32_t)((n - row0 < seg_rows) ? (n - row0) : seg_rows) : 0u;
    const uint32_t q0 = qg * QG;
    v.nq = (b - q0 < (uint32_t)QG) ? (b - q0) : QG;
    v.qb = queries + (uint64_t)qg * ldq * QG;
  }
  __device__ __forceinline__ uint32_t seq_base(uint32_t it, int) const { return (it % n_segs_pad) * seg_rows; }
  __device__ __forceinline__ const uint32_t* seq_ids(uint32_t it) const { return ids + (uint64_t)(it % n_segs_pad) * seg_rows; }
  __device__ __forceinline__ uint64_t* out(uint32_t it, int qi) const {
    const uint32_t seg = it % n_segs_pad, qg = it / n_segs_pad;
    return partials + ((uint64_t)(qg * QG + qi) * n_segs + seg) * k;
  }
  __device__ __forceinline__ uint32_t bound_slot(uint32_t it, int qi) const { return (it / n_segs_pad) * QG + qi; }
};

// one block per query: the k smallest keys of its n_segs partial slots -> (id, dist) at ranks rank0 .. rank0 + k - 1 of output
// row q (pitch top_k).  top_k > 64 comes 64 ranks per pass (ScanParams::lower): this pass's last key is the next one's bound.
// NW = waves of the block: four while a wave holds its share of the slots' heads in registers (4096 slots), sixteen beyond
template <int NW>
__global__ __launch_bounds__(kWave * NW) void flat_merge_kernel(const uint64_t* partials, uint32_t n_segs, uint32_t k, uint64_t n,
                                                                uint32_t top_k, uint32_t rank0, uint64_t* out_ids, float* out_dist,
                                                                uint32_t* out_count, uint64_t* lower_out) {
  __shared__ uint64_t sh[NW][kWave];
  const uint32_t q = blockIdx.x;
  uint64_t list = block_merge_keys<NW>(partials + (uint64_t)q * n_segs * k, n_segs * k, k, sh);
  if (threadIdx.x >= kWave) return;
  const int lane = threadIdx.x;
  const uint32_t total = n < top_k ? (uint32_t)n : top_k;  // utils.rs:79 take(k) of n sorted rows
  if (lane < (int)k && rank0 + (uint32_t)lane < total) {
    out_ids[(uint64_t)q * top_k + rank0 + lane] = (uint32_t)list;
    out_dist[(uint64_t)q * top_k + rank0 + lane] = __uint_as_float(order_bits_to_f32_bits((uint32_t)(list >> 32)));
  }
  if (lower_out != nullptr && lane == (int)k - 1) lower_out[q] = list;  // (kKeyMax when the rows ran out: the next pass finds nothing)
  if (lane == 0 && rank0 == 0) out_count[q] = total;
}

}  // namespace vers

using namespace vers;

struct vers_flat {
  int device = 0;
  uint32_t d = 0, ld = 0;  // ld = round_up(d, kColAlign): columns of the blocked corpus and of padded queries
  uint64_t n = 0;
  float* rows = nullptr;   // lane-transposed tiles (scan.hip.h)
  size_t rows_cap = 0;
  int n_cu = 256;
  // workspace (grown on demand, never inside a steady-state call)
  float* q_stage = nullptr;
  size_t q_stage_cap = 0;
  float* q_up = nullptr;  // host-pointer calls: uploaded queries (grow-only: no allocation in a steady-state call)
  size_t q_up_cap = 0;
  float* zero_q = nullptr;
  uint32_t zero_q_len = 0;
  uint64_t* partials = nullptr;  // partial slots, then one pruning bound per query
  size_t partials_cap = 0;
  uint64_t* lower = nullptr;     // top_k > 64: the previous pass's last key per query
  size_t lower_cap = 0;
  size_t bounds_off = 0;
  uint32_t* status_dev = nullptr;
  uint64_t* o_ids = nullptr;
  float* o_dist = nullptr;
  uint32_t* o_cnt = nullptr;
  size_t o_cap = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_valid = false;
  FlatShadow shadow;  // fp16 shadow of the rows + what a single query's exact finish needs (flat_shadow.hpp); empty when it did not fit
  std::mutex mu;
};

namespace {

template <int QG, int METRIC>
int32_t launch_flat_scan(vers_flat* h, const FlatSrc<QG, false>& src, uint32_t n_items, hipStream_t st, const uint64_t* lower) {
  ScanParams p;
  p.ld = h->ld;
  p.n_chunks = h->ld / kChunk;
  p.k = src.k;
  p.status = h->status_dev;
  p.debug = scan_debug_flags();
  p.stamps = nullptr;
  p.next_quad = nullptr;
  p.bounds = nullptr;  // every item of a query runs concurrently here: a shared bound prunes nothing and its atomics contend
  p.lower = lower;     // (top_k > 64: ranks 64p .. 64p + 63 in pass p)
  const size_t lds = scan_lds_bytes(QG, h->ld);
  if (int32_t rc = scan_prepare_launch(scan_kernel<QG, METRIC, FlatSrc<QG, false>>, lds)) return rc;
  const uint32_t max_blocks = (uint32_t)h->n_cu * scan_blocks_per_cu(QG, h->ld);
  uint32_t blocks = (n_items + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks == 0) blocks = 1;
  if (lower == nullptr) VERS_HIP_TRY(hipEventRecord(h->ev0, st));  // (the measurement hook times the first pass)
  hipLaunchKernelGGL((scan_kernel<QG, METRIC, FlatSrc<QG, false>>), dim3(blocks), dim3(kWave * kWavesPerBlock), lds, st, src, p);
  VERS_HIP_TRY(hipGetLastError());
  if (lower == nullptr) VERS_HIP_TRY(hipEventRecord(h->ev1, st));
  h->ev_valid = true;
  return VERS_OK;
}

int32_t flat_search_dev_locked(vers_flat* h, const float* q_dev, uint64_t ldq, uint32_t b, uint32_t top_k,
                               uint32_t metric, uint64_t* out_ids, float* out_dist, uint32_t* out_count,
                               hipStream_t st) {
  if (b == 0) return VERS_OK;
  if (top_k == 0) {
    VERS_HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(uint32_t) * b, st));
    return VERS_OK;
  }
  const uint32_t ldq_pad = h->ld;
  const int QG = b == 1 ? 1 : 8;
  const uint32_t n_qg = (b + QG - 1) / QG;
  // queries: a single query is used in place when it is chunk-padded by construction; otherwise
  // stage a zero padded (QG > 1: interleaved) copy
  const float* q = q_dev;
  uint32_t ldq_use = (uint32_t)ldq;
  if (!(QG == 1 && h->d == h->ld)) {
    const uint64_t tot = (uint64_t)n_qg * ldq_pad * QG;
    if (int32_t rc = grow(h->q_stage, h->q_stage_cap, (size_t)tot)) return rc;
    if (int32_t rc = launch_stage_queries(q_dev, ldq, h->d, h->q_stage, ldq_pad, b, (uint32_t)QG, st)) return rc;
    q = h->q_stage;
    ldq_use = ldq_pad;
  }
  // A single query streams the fp16 shadow when there is one (round 5: half the bytes; pre-selection, certificate, exact re-score
  // and -- when the certificate fails -- the exact re-scan, like the inverted lists' single query; same results)
  if (b == 1 && flat_shadow_usable(h->shadow, h->n, h->ld, top_k)) {
    if ((reinterpret_cast<uintptr_t>(q) & 15u) != 0) {  // (the kernels read the query 16 bytes at a time)
      if (int32_t rc = grow(h->q_stage, h->q_stage_cap, (size_t)ldq_pad)) return rc;
      if (int32_t rc = launch_stage_queries(q_dev, ldq, h->d, h->q_stage, ldq_pad, 1, 1, st)) return rc;
      q = h->q_stage;
    }
    if (int32_t rc = flat_shadow_search1(h->shadow, h->rows, h->n, h->ld, h->n_cu, q, top_k, metric, h->status_dev, out_ids, out_dist, out_count, st, h->ev0, h->ev1)) return rc;
    h->ev_valid = true;
    return VERS_OK;
  }
  // one item per resident wave when possible (static balance), at least one 64-row tile each
  const uint32_t target_items = (uint32_t)h->n_cu * scan_blocks_per_cu(QG, h->ld) * kWavesPerBlock;
  uint64_t per = (h->n * n_qg + target_items - 1) / target_items;
  uint32_t seg_rows = (uint32_t)std::min<uint64_t>(round_up64(per ? per : 1, kWave), max_seg_rows(h->ld));
  uint32_t n_segs = (uint32_t)((h->n + seg_rows - 1) / seg_rows);
  if (n_segs == 0) n_segs = 1;
  // one key per lane is the width of every list in the kernels: wider results come 64 ranks per pass -- keys are unique and
  // totally ordered, pass p holds exactly ranks 64p .. 64p + 63, selected by an exclusive lower bound (utils.rs:68-82 has no cap)
  const uint32_t k_w = std::min<uint32_t>(top_k, kMaxTopK);
  h->bounds_off = (size_t)b * n_segs * k_w;
  if (int32_t rc = grow(h->partials, h->partials_cap, h->bounds_off + (size_t)n_qg * QG)) return rc;
  VERS_HIP_TRY(hipMemsetAsync(h->partials + h->bounds_off, 0xFF, (size_t)n_qg * QG * sizeof(uint64_t), st));
  if (top_k > (uint32_t)kMaxTopK)
    if (int32_t rc = grow(h->lower, h->lower_cap, (size_t)b)) return rc;
  const uint32_t n_segs_pad = QG == 1 ? n_segs : round_up(n_segs, 4);
  const uint32_t n_items = n_segs_pad * n_qg;

  // (a pass past the last row finds nothing: top_k = 100000 on n = 1000 is 16 passes, not 1563; 64-bit rank: no wrap near 2^32)
  for (uint64_t rank0 = 0; rank0 < top_k && (rank0 == 0 || rank0 < h->n); rank0 += kMaxTopK) {
    const uint32_t k_pass = (uint32_t)std::min<uint64_t>(kMaxTopK, top_k - rank0);
    const uint64_t* lower = rank0 ? h->lower : nullptr;
    auto fill = [&](auto& src) {
      src.rows = h->rows; src.n = h->n; src.ld = h->ld; src.seg_rows = seg_rows; src.n_segs = n_segs; src.n_segs_pad = n_segs_pad;
      src.queries = q; src.ldq = ldq_use; src.b = b; src.partials = h->partials;
      src.k = k_pass; src.ids = nullptr;
    };
    int32_t rc;
    if (QG == 1) {
      FlatSrc<1, false> src; fill(src);
      rc = metric == VERS_METRIC_L2SQ ? launch_flat_scan<1, 0>(h, src, n_items, st, lower) : launch_flat_scan<1, 1>(h, src, n_items, st, lower);
    } else {
      FlatSrc<8, false> src; fill(src);
      rc = metric == VERS_METRIC_L2SQ ? launch_flat_scan<8, 0>(h, src, n_items, st, lower) : launch_flat_scan<8, 1>(h, src, n_items, st, lower);
    }
    if (rc) return rc;
    uint64_t* lower_out = top_k > (uint32_t)kMaxTopK ? h->lower : (uint64_t*)nullptr;
    if (n_segs <= 4096) hipLaunchKernelGGL(flat_merge_kernel<4>, dim3(b), dim3(kWave * 4), 0, st, h->partials, n_segs, k_pass, h->n, top_k, (uint32_t)rank0, out_ids,
                                           out_dist, out_count, lower_out);
    else hipLaunchKernelGGL(flat_merge_kernel<kMergeWaves>, dim3(b), dim3(kWave * kMergeWaves), 0, st, h->partials, n_segs, k_pass, h->n, top_k, (uint32_t)rank0,
                            out_ids, out_dist, out_count, lower_out);
    VERS_HIP_TRY(hipGetLastError());
  }
  return VERS_OK;
}

int32_t check_args(vers_flat* h, uint32_t top_k, uint32_t metric) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  if (metric > VERS_METRIC_COSDIST) return fail(VERS_ERR_INVALID, "unknown metric");
  if (h->n > 0xFFFFFFFFull) return fail(VERS_ERR_INVALID, "more than 2^32-1 rows per handle");
  return VERS_OK;
}

}  // namespace

extern "C" {

int32_t vers_flat_create(int32_t device, uint32_t d, vers_flat_t** out) {
  if (!out || d == 0) return fail(VERS_ERR_INVALID, "vers_flat_create: bad arguments");
  int cnt = 0;
  VERS_HIP_TRY(hipGetDeviceCount(&cnt));
  if (device < 0 || device >= cnt) return fail(VERS_ERR_INVALID, "vers_flat_create: no such device");
  DeviceGuard g(device);
  vers_flat* h = new (std::nothrow) vers_flat();
  if (!h) return fail(VERS_ERR_INVALID, "out of host memory");
  h->device = device;
  h->d = d;
  h->ld = round_up(d, kColAlign);
  hipDeviceProp_t prop;
  VERS_HIP_TRY(hipGetDeviceProperties(&prop, device));
  h->n_cu = prop.multiProcessorCount;
  h->zero_q_len = h->ld;
  VERS_HIP_TRY(hipMalloc((void**)&h->zero_q, h->zero_q_len * sizeof(float)));
  VERS_HIP_TRY(hipMemset(h->zero_q, 0, h->zero_q_len * sizeof(float)));
  VERS_HIP_TRY(hipMalloc((void**)&h->status_dev, 16));
  VERS_HIP_TRY(hipMemset(h->status_dev, 0, 16));
  VERS_HIP_TRY(hipEventCreate(&h->ev0));
  VERS_HIP_TRY(hipEventCreate(&h->ev1));
  *out = h;
  return VERS_OK;
}

int32_t vers_flat_destroy(vers_flat_t* h) {
  if (!h) return VERS_OK;
  DeviceGuard g(h->device);
  (void)hipDeviceSynchronize();
  if (h->rows) (void)hipFree(h->rows);
  h->shadow.release();
  for (void* p : {(void*)h->q_stage, (void*)h->q_up, (void*)h->zero_q, (void*)h->partials, (void*)h->lower, (void*)h->status_dev, (void*)h->o_ids,
                  (void*)h->o_dist, (void*)h->o_cnt})
    if (p) (void)hipFree(p);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
  return VERS_OK;
}

int32_t vers_flat_upload(vers_flat_t* h, const float* rows, uint64_t n, uint64_t row_stride_bytes) {
  if (!h || (n && !rows) || row_stride_bytes < (uint64_t)h->d * 4 || row_stride_bytes % 4)
    return fail(VERS_ERR_INVALID, "vers_flat_upload: bad arguments");
  if (n > 0xFFFFFFFFull) return fail(VERS_ERR_INVALID, "more than 2^32-1 rows per handle");
  std::lock_guard<std::mutex> lk(h->mu);
  DeviceGuard g(h->device);
  h->n = 0;
  h->shadow.release();
  if (int32_t rc = grow(h->rows, h->rows_cap, (size_t)std::max<uint64_t>(1, blocked_floats(n, h->ld)))) return rc;
  if (n == 0) return VERS_OK;
  float* tmp = nullptr;  // row-major staging, re-laid out on the device
  VERS_HIP_TRY(hipMalloc((void**)&tmp, n * (size_t)h->d * sizeof(float)));
  int32_t rc = VERS_OK;
  if (hipMemcpy2D(tmp, (size_t)h->d * 4, rows, row_stride_bytes, (size_t)h->d * 4, n, hipMemcpyHostToDevice) != hipSuccess)
    rc = fail(VERS_ERR_HIP, "corpus upload failed");
  if (!rc) rc = launch_to_blocked(tmp, h->d, h->d, n, h->rows, h->ld, nullptr);
  (void)hipDeviceSynchronize();
  (void)hipFree(tmp);
  if (!rc) h->n = n;
  if (!rc) rc = flat_shadow_derive(h->shadow, h->rows, n, h->ld, h->n_cu);
  return rc;
}

int32_t vers_flat_upload_dev(vers_flat_t* h, const float* rows_dev, uint64_t n, uint64_t ld_floats) {
  if (!h || (n && !rows_dev) || ld_floats < h->d) return fail(VERS_ERR_INVALID, "vers_flat_upload_dev: bad arguments");
  if (n > 0xFFFFFFFFull) return fail(VERS_ERR_INVALID, "more than 2^32-1 rows per handle");
  std::lock_guard<std::mutex> lk(h->mu);
  DeviceGuard g(h->device);
  h->n = 0;
  h->shadow.release();
  if (int32_t rc = grow(h->rows, h->rows_cap, (size_t)std::max<uint64_t>(1, blocked_floats(n, h->ld)))) return rc;
  if (int32_t rc = launch_to_blocked(rows_dev, ld_floats, h->d, n, h->rows, h->ld, nullptr)) return rc;
  VERS_HIP_TRY(hipDeviceSynchronize());
  h->n = n;
  return flat_shadow_derive(h->shadow, h->rows, n, h->ld, h->n_cu);
}

int32_t vers_flat_search_dev(vers_flat_t* h, const float* queries_dev, uint64_t ldq_floats, uint32_t b, uint32_t top_k,
                             uint32_t metric, uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_count_dev,
                             void* stream) {
  if (int32_t rc = check_args(h, top_k, metric)) return rc;
  if (b && (!queries_dev || ldq_floats < h->d || !out_count_dev || (top_k && (!out_ids_dev || !out_dist_dev))))
    return fail(VERS_ERR_INVALID, "vers_flat_search_dev: bad arguments");
  std::lock_guard<std::mutex> lk(h->mu);
  DeviceGuard g(h->device);
  return flat_search_dev_locked(h, queries_dev, ldq_floats, b, top_k, metric, out_ids_dev, out_dist_dev, out_count_dev,
                                (hipStream_t)stream);
}

int32_t vers_flat_poll(vers_flat_t* h, void* stream) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  DeviceGuard g(h->device);
  uint32_t st = 0;
  VERS_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  VERS_HIP_TRY(hipMemcpy(&st, h->status_dev, sizeof(st), hipMemcpyDeviceToHost));
  if (st) {
    VERS_HIP_TRY(hipMemset(h->status_dev, 0, sizeof(st)));
    if (st & 1u) return fail(VERS_ERR_NAN, "NaN distance (the reference panics in partial_cmp().unwrap())");
  }
  return VERS_OK;
}

int32_t vers_flat_search(vers_flat_t* h, const float* queries, uint64_t q_stride_bytes, uint32_t b, uint32_t top_k,
                         uint32_t metric, uint64_t* out_ids, float* out_dist, uint32_t* out_count) {
  if (int32_t rc = check_args(h, top_k, metric)) return rc;
  if (b && (!queries || q_stride_bytes < (uint64_t)h->d * 4 || q_stride_bytes % 4 || !out_count ||
            (top_k && (!out_ids || !out_dist))))
    return fail(VERS_ERR_INVALID, "vers_flat_search: bad arguments");
  if (b == 0) return VERS_OK;
  std::lock_guard<std::mutex> lk(h->mu);
  DeviceGuard g(h->device);
  const uint32_t ldq_pad = h->ld;
  if (int32_t rc0 = grow(h->q_up, h->q_up_cap, (size_t)b * ldq_pad)) return rc0;
  float* qd = h->q_up;
  int32_t rc = VERS_OK;
  do {
    if (hipMemset(qd, 0, (size_t)b * ldq_pad * sizeof(float)) != hipSuccess ||
        hipMemcpy2D(qd, (size_t)ldq_pad * 4, queries, q_stride_bytes, (size_t)h->d * 4, b, hipMemcpyHostToDevice) != hipSuccess) {
      rc = fail(VERS_ERR_HIP, "query upload failed");
      break;
    }
    const size_t need = (size_t)b * (top_k ? top_k : 1);
    if (need > h->o_cap) {
      size_t c0 = h->o_cap, c1 = h->o_cap, c2 = h->o_cap;
      if ((rc = grow(h->o_ids, c0, need)) || (rc = grow(h->o_dist, c1, need)) || (rc = grow(h->o_cnt, c2, need))) break;
      h->o_cap = need;
    }
    rc = flat_search_dev_locked(h, qd, ldq_pad, b, top_k, metric, h->o_ids, h->o_dist, h->o_cnt, nullptr);
    if (rc) break;
    rc = vers_flat_poll(h, nullptr);
    if (rc) break;
    if (top_k) {
      if (hipMemcpy(out_ids, h->o_ids, need * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess ||
          hipMemcpy(out_dist, h->o_dist, need * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) {
        rc = fail(VERS_ERR_HIP, "result download failed");
        break;
      }
    }
    if (hipMemcpy(out_count, h->o_cnt, b * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
      rc = fail(VERS_ERR_HIP, "result download failed");
  } while (0);
  return rc;
}

int32_t vers_flat_last_scan_ms(vers_flat_t* h, float* out_ms) {
  if (!h || !out_ms) return fail(VERS_ERR_INVALID, "bad arguments");
  if (!h->ev_valid) return fail(VERS_ERR_INVALID, "no scan has been launched on this handle");
  DeviceGuard g(h->device);
  VERS_HIP_TRY(hipEventSynchronize(h->ev1));
  VERS_HIP_TRY(hipEventElapsedTime(out_ms, h->ev0, h->ev1));
  return VERS_OK;
}

}  // extern "C"
