// kmeans.hip -- the k-means steps of IVFFlatIndex::build_index on the device
// (reference: /root/reference/vers/src/indexes/ivfflat.rs:29-100,138-149).
//
// Every step keeps the reference's arithmetic ORDER, so results are bit-identical:
//   assign : per (point, centroid) the sequential f32 sum of (c-x)^2 -- the scan engine with the
//            centroid matrix as the "corpus" (lane == centroid) and 8 points per wave as uniform
//            operands; first-minimum tie rule through the (dist, centroid index) key; top-1.
//   update : per (cluster, column) the sum over members in ascending point index, then / count.
//   cost   : strict left-to-right fold of the per-point minimum distances (one lane: the fold is
//            inherently serial; it reads 4 bytes per point and is ~1e-3 of an assign pass).
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <vector>

#include <chrono>
#include <map>
#include <mutex>
#include <vector>

#include "kmeans.hpp"
#include "gemm.hip.h"
#include "scan.hip.h"
#include "util.hip.h"

namespace vers {

// process-wide accounting of the device memory held through DevBuf (vers_mem_stats)
static std::atomic<uint64_t> g_dev_now{0}, g_dev_peak{0};
void dev_mem_account(int64_t delta) {
  const uint64_t now = g_dev_now.fetch_add((uint64_t)delta) + (uint64_t)delta;
  uint64_t pk = g_dev_peak.load();
  while (now > pk && !g_dev_peak.compare_exchange_weak(pk, now)) {}
}
void dev_mem_stats(uint64_t* now, uint64_t* peak, bool reset_peak) {
  if (now) *now = g_dev_now.load();
  if (peak) *peak = g_dev_peak.load();
  if (reset_peak) g_dev_peak = g_dev_now.load();
}

int32_t DevBuf::reserve(size_t bytes) {
  if (bytes <= cap) return VERS_OK;
  release();
  VERS_HIP_TRY(hipMalloc(&p, bytes));
  cap = bytes;
  dev_mem_account((int64_t)bytes);
  // diagnosis (option "poison_alloc" = a byte): every new device buffer starts filled with that byte instead of whatever the
  // allocator hands out -- nothing may depend on uninitialised scratch (the GPU suite passes under 0x7f, 0xa5 and 0x00)
  const int64_t pa = opt_get("poison_alloc", -1);
  const int poison = pa >= 0 ? (int)(pa & 0xFF) : -1;
  if (poison >= 0) {
    VERS_HIP_TRY(hipMemset(p, poison, bytes));
    VERS_HIP_TRY(hipDeviceSynchronize());
  }
  return VERS_OK;
}
void DevBuf::release() {
  if (p) {
    (void)hipFree(p);
    dev_mem_account(-(int64_t)cap);
  }
  p = nullptr;
  cap = 0;
}

// ---- assign -------------------------------------------------------------------------
// item = (group of QG points, quarter of the centroid matrix): a quad of items shares the points' query
// block (scan.hip.h), each wave keeps the running first-minimum over its quarter of the centroids.
template <int QG>
struct AssignSrc {
  static constexpr bool kSeqIds = false;
  static constexpr bool kStreamOnce = false;  // the centroid matrix is re-read by every point group
  const float* C;  // centroids, lane-transposed tiles
  uint32_t k, ld;
  uint32_t seg_rows;     // centroids per quarter (multiple of 64)
  const float* qblocks;  // [ceil(nb/QG)][ldq][QG]
  uint32_t ldq, nb;
  uint64_t* keys;  // [nb][4]: one key per (point, centroid quarter)
  __device__ __forceinline__ uint32_t n_items() const { return ((nb + QG - 1) / QG) * 4; }
  __device__ __forceinline__ void get(uint32_t it, ItemView<QG>& v) const {
    const uint32_t g = it >> 2, w = it & 3;
    const uint32_t r0 = w * seg_rows;
    v.rows = C + (uint64_t)(r0 < k ? r0 : 0) * ld;
    v.nrows = r0 < k ? (k - r0 < seg_rows ? k - r0 : seg_rows) : 0u;
    v.nq = (nb - g * QG < (uint32_t)QG) ? (nb - g * QG) : QG;
    v.qb = qblocks + (uint64_t)g * ldq * QG;
  }
  __device__ __forceinline__ uint32_t seq_base(uint32_t it, int) const { return (it & 3) * seg_rows; }
  __device__ __forceinline__ const uint32_t* seq_ids(uint32_t) const { return nullptr; }
  __device__ __forceinline__ uint64_t* out(uint32_t it, int qi) const { return keys + ((uint64_t)(it >> 2) * QG + qi) * 4 + (it & 3); }
  __device__ __forceinline__ uint32_t bound_slot(uint32_t, int) const { return 0; }
};

// first minimum over the (up to) four quarter keys of each point: keys order = (distance, centroid index)
__global__ void keys_to_assign_kernel(const uint64_t* keys, uint32_t nb, uint32_t* assign, float* mind) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb) return;
  uint64_t key = keys[(uint64_t)i * 4];
#pragma unroll
  for (int w = 1; w < 4; ++w) {
    const uint64_t kw = keys[(uint64_t)i * 4 + w];
    key = kw < key ? kw : key;
  }
  assign[i] = (uint32_t)key;
  if (mind) mind[i] = __uint_as_float(order_bits_to_f32_bits((uint32_t)(key >> 32)));
}

int32_t km_assign(const float* X, uint32_t ldx, uint64_t n, const float* C, uint32_t ldc, uint32_t k, uint32_t d,
                  uint32_t* out_assign, float* out_mind, KMeansScratch& ws, int n_cu, hipStream_t st, int metric) {
  constexpr int QG = 8;
  if (n == 0) return VERS_OK;
  const uint32_t ldq = round_up(d, kColAlign);  // columns of the blocked centroids == padded point length
  if (int32_t rc = ws.cblocked.reserve(blocked_floats(k, ldq) * sizeof(float))) return rc;
  if (int32_t rc = launch_to_blocked(C, ldc, d, k, ws.cblocked.as<float>(), ldq, st)) return rc;
  // batch so that the interleaved staging stays ~<= 1 GiB
  uint64_t batch = (1ull << 30) / ((uint64_t)ldq * 4);
  batch = batch / QG * QG;
  if (batch < (uint64_t)QG * 1024) batch = (uint64_t)QG * 1024;
  if (batch > n) batch = round_up64(n, QG);
  if (int32_t rc = ws.qblocks.reserve(batch * ldq * sizeof(float))) return rc;
  if (int32_t rc = ws.keys.reserve(batch * 4 * sizeof(uint64_t))) return rc;
  if (int32_t rc = ws.status.reserve(16)) return rc;
  ScanParams p;
  p.ld = ldq;
  p.n_chunks = ldq / kChunk;
  p.k = 1;
  p.status = ws.status.as<uint32_t>();
  p.debug = 0;
  p.stamps = nullptr;
  p.next_quad = nullptr;
  p.bounds = nullptr; p.lower = nullptr;  // the four quarter-items of a point group run side by side: nothing to share
  const size_t lds = scan_lds_bytes(QG, ldq);
  if (int32_t rc = metric ? scan_prepare_launch(scan_kernel<QG, 1, AssignSrc<QG>>, lds) : scan_prepare_launch(scan_kernel<QG, 0, AssignSrc<QG>>, lds)) return rc;
  const uint32_t seg_rows = round_up((k + 3) / 4, 64);
  for (uint64_t i0 = 0; i0 < n; i0 += batch) {
    const uint32_t nb = (uint32_t)((n - i0 < batch) ? (n - i0) : batch);
    if (int32_t rc = launch_stage_queries(X + i0 * ldx, ldx, d, ws.qblocks.as<float>(), ldq, nb, QG, st)) return rc;
    AssignSrc<QG> src;
    src.C = ws.cblocked.as<float>(); src.k = k; src.ld = ldq; src.seg_rows = seg_rows; src.qblocks = ws.qblocks.as<float>();
    src.ldq = ldq; src.nb = nb; src.keys = ws.keys.as<uint64_t>();
    VERS_HIP_TRY(hipMemsetAsync(ws.keys.p, 0xFF, (size_t)nb * 4 * sizeof(uint64_t), st));  // empty quarters stay "no candidate"
    uint32_t blocks = (nb + QG - 1) / QG;  // one block per quad of items
    const uint32_t max_blocks = (uint32_t)n_cu * scan_blocks_per_cu(QG, ldq);
    if (blocks > max_blocks) blocks = max_blocks;
    if (metric) hipLaunchKernelGGL((scan_kernel<QG, 1, AssignSrc<QG>>), dim3(blocks), dim3(kWave * kWavesPerBlock), lds, st, src, p);
    else hipLaunchKernelGGL((scan_kernel<QG, 0, AssignSrc<QG>>), dim3(blocks), dim3(kWave * kWavesPerBlock), lds, st, src, p);
    VERS_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(keys_to_assign_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, ws.keys.as<uint64_t>(), nb,
                       out_assign + i0, out_mind ? out_mind + i0 : nullptr);
    VERS_HIP_TRY(hipGetLastError());
  }
  return VERS_OK;
}

// ---- assign through the matrix cores ------------------------------------------------------
static std::atomic<uint64_t> g_mfma_points{0}, g_mfma_fallbacks{0};

// ---- build timing (vers_build_stats) -------------------------------------------------------------------------------
static std::mutex g_bs_mu;
static BuildStats g_bs;  // (guarded by g_bs_mu)
BuildStats build_stats() {
  std::lock_guard<std::mutex> lk(g_bs_mu);
  return g_bs;
}
void build_stats_add(double BuildStats::*field, double v) {
  std::lock_guard<std::mutex> lk(g_bs_mu);
  g_bs.*field += v;
}
namespace {
struct Stretch { hipEvent_t a = nullptr, b = nullptr; double BuildStats::*field = nullptr; };
std::vector<Stretch> g_stretches;   // closed stretches whose events have not been read yet (guarded by g_bs_mu)
std::vector<hipEvent_t> g_ev_pool;  // (guarded by g_bs_mu)
hipEvent_t ev_get() {
  {
    std::lock_guard<std::mutex> lk(g_bs_mu);
    if (!g_ev_pool.empty()) { hipEvent_t e = g_ev_pool.back(); g_ev_pool.pop_back(); return e; }
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return e;
}
}  // namespace
KmTimer::KmTimer(hipStream_t stream, double BuildStats::*f) : st(stream), field(f), a(ev_get()), b(ev_get()) {
  if (a) (void)hipEventRecord(a, st);
}
KmTimer::~KmTimer() {
  if (b) (void)hipEventRecord(b, st);
  std::lock_guard<std::mutex> lk(g_bs_mu);
  Stretch s; s.a = a; s.b = b; s.field = field;
  g_stretches.push_back(s);
}
void km_timers_collect() {
  std::lock_guard<std::mutex> lk(g_bs_mu);
  std::vector<Stretch> keep;
  for (auto& s : g_stretches) {
    // (another thread's build may have closed a stretch its stream has not reached yet: it stays for that build's collect)
    if (s.b && hipEventQuery(s.b) == hipErrorNotReady) { keep.push_back(s); continue; }
    (void)hipGetLastError();
    float ms = 0.0f;
    if (s.a && s.b && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { if (s.field) g_bs.*(s.field) += ms; }
    else (void)hipGetLastError();
    if (s.a) g_ev_pool.push_back(s.a);
    if (s.b) g_ev_pool.push_back(s.b);
  }
  g_stretches.swap(keep);
}

static std::atomic<int> g_x3_mask{-1};
int gemm_x3_mask() {
  int m = g_x3_mask.load();
  if (m < 0) {  // default: both contractions as bf16x3 (build 3.3 -> 2.1 s, coarse GEMM 62 -> ~25 us, same bits)
    m = (int)opt_get("gemm_x3", 3) & 3;
    g_x3_mask = m;
  }
  return m;
}
void set_gemm_x3_mask(int m) { g_x3_mask = m & 3; }

bool km_use_mfma(uint64_t n, uint32_t k, uint32_t d) {
  const int mode = (int)opt_get("assign", 0);
  if (k < 2 || n == 0) return false;
  if (mode == 1) return false;
  if (mode == 2) return true;
  return (double)n * k * d >= 1e11;  // below this the exact scan is a few ms and the extra launches do not pay
}

int32_t km_assign_mfma(const float* X, uint32_t ldx, uint64_t n, const float* C, uint32_t ldc, uint32_t k, uint32_t d,
                       uint32_t* out_assign, float* out_mind, KMeansScratch& ws, int n_cu, hipStream_t st, int metric) {
  if (n == 0) return VERS_OK;
  const auto wall0 = std::chrono::steady_clock::now();  // (the pass from its first allocation: the first one of a process pays the cold start)
  const uint32_t ldq = round_up(d, kColAlign);
  const uint32_t k_pad = round_up(k, kGemmBM);
  if (int32_t rc = ws.cg.reserve((size_t)k_pad * ldq * sizeof(float))) return rc;
  if (int32_t rc = ws.cnorm.reserve(((size_t)k_pad + 4) * sizeof(float))) return rc;
  if (int32_t rc = ws.status.reserve(16)) return rc;
  if (int32_t rc = ws.fb.reserve((n + 4) * sizeof(uint32_t))) return rc;
  uint32_t* fb_list = ws.fb.as<uint32_t>();
  uint32_t* fb_count = fb_list + n;
  float* cmax2_dev = ws.cnorm.as<float>() + k_pad;
  VERS_HIP_TRY(hipMemsetAsync(ws.cg.p, 0, (size_t)k_pad * ldq * sizeof(float), st));
  VERS_HIP_TRY(hipMemsetAsync(fb_count, 0, sizeof(uint32_t), st));
  // large k: uncertified points first go through the tile-limited re-scan (assign_tile_rescan_kernel); what it cannot settle
  // lands in fb_list like before
  // THE CASCADE (round 6; option "assign_terms": 0 auto, 1 always, 3 never): the wide contraction with ONE product of FP16 operands --
  // a third of the MFMAs, half the LDS traffic -- is the first filter; its certificate is as wide as the operands' MEASURED fp16 residuals
  // make it (gemm.hip.h) and the points it leaves open go to the tile-limited exact re-scan below.  Whether that pays depends on the data (clustered rows beat
  // their runner-up centroid by far more than the window; rows spread evenly over the sphere do not): in auto mode a pass starts with
  // a small probing batch and keeps the cascade while at most 1/8 of a batch's points stay open; the verdict is remembered in the
  // scratch for the following passes over the same (n, k).
  const int terms_opt = (int)opt_get("assign_terms", 0);
  const bool wide_shape = (gemm_x3_mask() & 1) != 0 && round_up(k, kGemmBM) % kGemmWide == 0;
  if (ws.cascade_n != n || ws.cascade_k != k) { ws.cascade_n = n; ws.cascade_k = k; ws.cascade = -1; }
  bool hi_only = wide_shape && terms_opt != 3 && (terms_opt == 1 || ws.cascade != 0);
  const bool probing = hi_only && terms_opt == 0 && ws.cascade < 0;
  const bool tiles_on = opt_get("assign_tiles", 1) != 0 || hi_only;
  // (from 64 tiles = k >= 8192 on: measured at k = 4096, N = 4M with option "assign_tiles_min" = 8 the pass gets 3 % SLOWER, 104.6 vs
  // 101.1 ms -- a launch of 2048 waves per batch against one exact scan of the 1.6 % uncertified points at the end; same bits)
  const uint32_t tiles_min = (uint32_t)opt_get("assign_tiles_min", 64);
  const bool tile_rescan = tiles_on && (k_pad / kGemmBM >= tiles_min || hi_only);
  uint32_t* fbq_list = nullptr; uint32_t* fbq_count = nullptr; float* fbq_thr = nullptr;
  if (tile_rescan) {
    // the centroids in the scan layout (lane-transposed 64-row tiles): what the re-scan streams
    if (int32_t rc = ws.cblocked.reserve(blocked_floats(k, ldq) * sizeof(float))) return rc;
    if (int32_t rc = launch_to_blocked(C, ldc, d, k, ws.cblocked.as<float>(), ldq, st)) return rc;
    if (int32_t rc = ws.fbq.reserve((2 * n + 4) * sizeof(uint32_t))) return rc;
    fbq_list = ws.fbq.as<uint32_t>(); fbq_count = fbq_list + n; fbq_thr = ws.fbq.as<float>() + n + 4;
    VERS_HIP_TRY(hipMemsetAsync(fbq_count, 0, 2 * sizeof(uint32_t), st));  // (the count and, behind it, where the current batch's entries begin)
  }
  if (int32_t rc = launch_stage_queries(C, ldc, d, ws.cg.as<float>(), ldq, k, 1, st)) return rc;
  hipLaunchKernelGGL(row_norms_kernel, dim3((k_pad + 255) / 256), dim3(256), 0, st, ws.cg.as<float>(), ldq, k, k_pad, ws.cnorm.as<float>());
  hipLaunchKernelGGL(max_norm_kernel, dim3(1), dim3(256), 0, st, ws.cnorm.as<float>(), k, cmax2_dev);
  VERS_HIP_TRY(hipGetLastError());
  const __bf16 *cg_h = nullptr, *cg_l = nullptr;
  if (gemm_x3_mask() & 1) {  // every block of the pass re-reads the centroids: split them into bf16 hi / lo once, not per tile
    const size_t ne = (size_t)k_pad * ldq;
    if (int32_t rc = ws.cg_s.reserve(2 * ne * sizeof(uint16_t))) return rc;
    VERS_HIP_TRY(launch_split_bf16(ws.cg.as<float>(), ne, ws.cg_s.as<__bf16>(), ws.cg_s.as<__bf16>() + ne, st));
    cg_h = ws.cg_s.as<__bf16>(); cg_l = cg_h + ne;
  }
  const __bf16* cg_f16 = nullptr;  // the centroids as fp16 (the cascade's operand) + the largest squared residual of a centroid row
  uint32_t* rc2_bits = nullptr;
  if (hi_only) {
    const size_t ne = (size_t)k_pad * ldq;
    if (int32_t rc = ws.cg_f16.reserve(ne * sizeof(uint16_t) + 16)) return rc;
    rc2_bits = reinterpret_cast<uint32_t*>(ws.cg_f16.as<uint16_t>() + ne);
    VERS_HIP_TRY(hipMemsetAsync(rc2_bits, 0, 16, st));
    hipLaunchKernelGGL(to_f16_resid_kernel, dim3(k_pad), dim3(256), 0, st, (const float*)ws.cg.as<float>(), ldq, k_pad, ws.cg_f16.as<_Float16>(), rc2_bits);
    VERS_HIP_TRY(hipGetLastError());
    cg_f16 = ws.cg_f16.as<__bf16>();
  }
  // point batch: the per-(centroid tile, point) triples stay <= 1 GiB (the GEMM never writes its product)
  const uint32_t n_tiles = k_pad / kGemmBM;
  // (batches in whole 256-point tiles when the wide contraction kernel can run: gemm_wide_ok)
  const bool wide = gemm_wide_ok(k_pad, kGemmWide, cg_h != nullptr);
  const uint64_t bn = wide ? (uint64_t)kGemmWide : (uint64_t)kGemmBN;
  uint64_t mb = ((1ull << 30) / ((uint64_t)n_tiles * 12)) / bn * bn;
  if (mb > 131072) mb = 131072;
  if (mb < bn) mb = bn;
  if (mb > round_up64(n, bn)) mb = round_up64(n, bn);
  if (int32_t rc = ws.gt.reserve((size_t)n_tiles * mb * 12)) return rc;
  float* part_v1 = ws.gt.as<float>();
  uint32_t* part_c1 = ws.gt.as<uint32_t>() + (size_t)n_tiles * mb;
  float* part_v2 = ws.gt.as<float>() + 2 * (size_t)n_tiles * mb;
  if (int32_t rc = ws.best.reserve(mb * 8)) return rc;
  uint32_t* best = ws.best.as<uint32_t>();
  float* g2 = ws.best.as<float>() + mb;
  const bool in_place_ok = ldx == ldq && d == ldq;  // (padding columns of the caller's X may hold anything: stage them away)
  uint32_t open_before = 0;
  for (uint64_t i0 = 0, step = 0; i0 < n; i0 += step) {
    // (a probing pass opens with a batch of at most 16384 points)
    const uint64_t want = (probing && i0 == 0) ? std::min<uint64_t>(mb, round_up64(16384, bn)) : mb;
    const uint32_t nb = (uint32_t)((n - i0 < want) ? (n - i0) : want);
    step = nb;
    const uint32_t nb_pad = round_up(nb, (uint32_t)bn);
    const float* xb = X + i0 * ldx;
    const float* xb_padded = in_place_ok ? xb : nullptr;  // this batch's rows with zeroed padding columns, pitch ldq (the tile re-scan's operand)
    // both operands fp16 in memory, staged by LDS-DMA (dist_gemm_h_kernel) from 4096 centroids on: the batch's conversion pass (0.11 ms per
    // 131072 x 768 points) is a tenth of the contraction at k = 4096 -- 1.03 ms with it against 1.04 for the register-staged kernel that
    // converts while it stages -- and nothing at k = 65536 (12.6 vs 16.2 ms).  Option "assign_glds": 1 always, 0 never.
    const int glds_opt = (int)opt_get("assign_glds", -1);
    const bool h_fits = (uint64_t)k_pad * ldq * 2u < (1ull << 32) && (uint64_t)mb * ldq * 2u < (1ull << 32);  // (its source offsets are 32 bits wide)
    const bool use_h = hi_only && wide && gemm_h_ok(ldq) && h_fits && (glds_opt > 0 || (glds_opt < 0 && k_pad >= 4096));
    if (use_h) {
      if (int32_t rc = ws.xh.reserve((size_t)mb * ldq * sizeof(uint16_t))) return rc;
    } else if (!in_place_ok || nb_pad != nb) {  // pad the columns / the tail rows through a staged copy
      if (int32_t rc = ws.xp.reserve((size_t)mb * ldq * sizeof(float))) return rc;
      if (nb_pad != nb) VERS_HIP_TRY(hipMemsetAsync(ws.xp.as<float>() + (size_t)nb * ldq, 0, (size_t)(nb_pad - nb) * ldq * sizeof(float), st));
      if (int32_t rc = launch_stage_queries(xb, ldx, d, ws.xp.as<float>(), ldq, nb, 1, st)) return rc;
      xb = ws.xp.as<float>();
      xb_padded = xb;
    }
    // (the triples are addressed with pitch mb: nb_pad <= mb)
    {
      KmTimer t(st, &BuildStats::gemm_ms);
      if (use_h) {  // (the conversion is part of the filter's price: inside the timer)
        const uint64_t work = (uint64_t)nb_pad * (ldq / 8);
        hipLaunchKernelGGL(rows_to_f16_pad_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, xb, (uint64_t)ldx, d, nb, nb_pad, ldq, ws.xh.as<_Float16>());
        VERS_HIP_TRY(launch_gemm_h(k_pad, nb_pad, st, ws.xh.as<_Float16>(), reinterpret_cast<const _Float16*>(cg_f16), ws.cnorm.as<float>(), ldq, (uint32_t)mb, metric, k,
                                   part_v1, part_c1, part_v2));
      } else if (wide)
        VERS_HIP_TRY(launch_gemm_wide(k_pad, nb_pad, st, xb, hi_only ? cg_f16 : cg_h, cg_l, ws.cnorm.as<float>(), ldq, (uint32_t)mb, metric, k, part_v1, part_c1, part_v2, hi_only));
      else
        VERS_HIP_TRY(launch_gemm<true>((gemm_x3_mask() & 1) != 0, k_pad / kGemmBM, nb_pad / kGemmBN, st, ws.cg.as<float>(), xb, ws.cnorm.as<float>(), ldq,
                                       (uint32_t)mb, (float*)nullptr, metric, k, part_v1, part_c1, part_v2, cg_h, cg_l));
    }
    build_stats_add(&BuildStats::gemm_launches, 1.0);
    build_stats_add(&BuildStats::gemm_flop, 2.0 * (double)nb * (double)k * (double)d);
    hipLaunchKernelGGL(assign_argmin_merge_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, (const float*)part_v1, (const uint32_t*)part_c1,
                       (const float*)part_v2, n_tiles, (uint32_t)mb, nb, best, g2, tile_rescan ? fbq_count + 1 : (uint32_t*)nullptr,
                       (const uint32_t*)fbq_count);
    hipLaunchKernelGGL(assign_rescore_kernel, dim3((nb + 63) / 64), dim3(64), 0, st, X + i0 * ldx, ldx, C, ldc, d, ldq, cmax2_dev, best, g2,
                       nb, k, (uint32_t)i0, out_assign + i0, out_mind ? out_mind + i0 : nullptr, tile_rescan ? fbq_list : fb_list,
                       tile_rescan ? fbq_count : fb_count, ws.status.as<uint32_t>(), metric, fbq_thr, hi_only && wide ? (const uint32_t*)rc2_bits : (const uint32_t*)nullptr);
    if (tile_rescan) {
      if (xb_padded == nullptr) {  // the re-scan reads this batch's rows zero padded to ldq columns (its scalar operand): staged when the caller's are not
        if (int32_t rc = ws.xp.reserve((size_t)mb * ldq * sizeof(float))) return rc;
        if (int32_t rc = launch_stage_queries(X + i0 * ldx, ldx, d, ws.xp.as<float>(), ldq, nb, 1, st)) return rc;
        xb_padded = ws.xp.as<float>();
      }
      hipLaunchKernelGGL(assign_tile_rescan_kernel, dim3(4096), dim3(kWave * kRescanWaves), 0, st, xb_padded, (const float*)ws.cblocked.as<float>(), ldq, k,
                         (const float*)part_v1, n_tiles, (uint32_t)mb, (uint32_t)i0, nb, (const uint32_t*)fbq_list, (const float*)fbq_thr,
                         (const uint32_t*)fbq_count, out_assign, out_mind, fb_list, fb_count, metric, (const uint32_t*)(fbq_count + 1), (const uint32_t*)part_c1,
                         (const float*)part_v2);
    }
    VERS_HIP_TRY(hipGetLastError());
    if (hi_only && wide && terms_opt == 0 && (i0 == 0 || ws.cascade < 0)) {  // the cascade's verdict: how many of this batch's points stayed open
      uint32_t open_now = 0;
      VERS_HIP_TRY(hipMemcpyAsync(&open_now, fbq_count, 4, hipMemcpyDeviceToHost, st));
      VERS_HIP_TRY(hipStreamSynchronize(st));
      const bool keep = (uint64_t)(open_now - open_before) * 8u <= nb;
      ws.cascade = keep ? 1 : 0;
      if (!keep) hi_only = false;  // (the rest of this pass and the following ones: three products)
      open_before = open_now;
    }
  }
  uint32_t nf = 0;
  VERS_HIP_TRY(hipMemcpyAsync(&nf, fb_count, 4, hipMemcpyDeviceToHost, st));
  VERS_HIP_TRY(hipStreamSynchronize(st));
  g_mfma_points += n;
  g_mfma_fallbacks += nf;
  if (nf) {  // uncertified points: the exact scan decides (order of fb_list is irrelevant: results are scattered by index)
    if (int32_t rc = ws.xf.reserve((size_t)nf * ldx * sizeof(float))) return rc;
    if (int32_t rc = ws.fa.reserve((size_t)nf * 4)) return rc;
    if (int32_t rc = ws.fm.reserve((size_t)nf * 4)) return rc;
    const uint64_t words = (uint64_t)nf * ldx;
    hipLaunchKernelGGL(gather_points_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, X, ldx, fb_list, nf, ws.xf.as<float>());
    VERS_HIP_TRY(hipGetLastError());
    if (int32_t rc = km_assign(ws.xf.as<float>(), ldx, nf, C, ldc, k, d, ws.fa.as<uint32_t>(), ws.fm.as<float>(), ws, n_cu, st, metric)) return rc;
    hipLaunchKernelGGL(scatter_assign_kernel, dim3((nf + 255) / 256), dim3(256), 0, st, fb_list, nf, ws.fa.as<uint32_t>(), ws.fm.as<float>(),
                       out_assign, out_mind);
    VERS_HIP_TRY(hipGetLastError());
    VERS_HIP_TRY(hipStreamSynchronize(st));
  }
  km_timers_collect();
  {
    const double pass_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    std::lock_guard<std::mutex> lk(g_bs_mu);
    if (g_bs.assign_passes == 0) g_bs.assign_first_ms = pass_ms;
    g_bs.assign_ms += pass_ms;
    g_bs.assign_passes += 1.0;
  }
  build_stats_add(&BuildStats::redone_points, (double)nf);
  return VERS_OK;
}

// ---- group ----------------------------------------------------------------------------
__global__ void count_kernel(const uint32_t* assign, uint32_t n, uint32_t* counts) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&counts[assign[i]], 1u);
}

// single block: starts[0..k] = exclusive prefix of counts
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(const uint32_t* counts, uint32_t k, uint32_t* starts) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < k; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < k ? counts[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
      uint32_t t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    const uint32_t incl = sh[threadIdx.x];
    if (i < k) starts[i] = carry + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) starts[k] = carry;
}

int32_t km_group(const uint32_t* assign, uint32_t n, uint32_t k, uint32_t* sorted_ids, uint32_t* counts, uint32_t* starts,
                 KMeansScratch& ws, hipStream_t st) {
  VERS_HIP_TRY(hipMemsetAsync(counts, 0, (size_t)k * sizeof(uint32_t), st));
  if (n) {
    hipLaunchKernelGGL(count_kernel, dim3((n + 255) / 256), dim3(256), 0, st, assign, n, counts);
    VERS_HIP_TRY(hipGetLastError());
  }
  hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, st, counts, k, starts);
  VERS_HIP_TRY(hipGetLastError());
  const size_t tb = group_by_cluster_temp_bytes(n, k);
  if (int32_t rc = ws.sort_tmp.reserve(tb)) return rc;
  return group_by_cluster(assign, n, k, sorted_ids, ws.sort_tmp.p, tb, st);
}

// ---- update -----------------------------------------------------------------------------
// thread = one (cluster, column); members visited in list order (ascending point index);
// 8 independent row loads in flight, the adds stay strictly ordered.  The sum CONTINUES from S[c][col]: zero for a
// single process (the reference starts from Vector([0.0; N]), ivfflat.rs:49), the previous rank's running sum when
// the rows are sharded over processes in ascending ranges -- the order of the additions is the reference's either way.
__global__ __launch_bounds__(256) void update_sums_kernel(const float* X, uint32_t ld, uint32_t d, const uint32_t* sorted_ids,
                                                          const uint32_t* starts, float* S, uint32_t ldc) {
  const uint32_t c = blockIdx.x;
  const uint32_t col = blockIdx.y * 256 + threadIdx.x;
  if (col >= ldc) return;
  if (col >= d) return;  // beyond the vector: the padding of S stays zero (the caller's padding columns of X may hold anything)
  const uint32_t s = starts[c], e = starts[c + 1];
  float acc = S[(uint64_t)c * ldc + col];
  uint32_t t = s;
  for (; t + 8 <= e; t += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = X[(uint64_t)sorted_ids[t + u] * ld + col];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __fadd_rn(acc, v[u]);
  }
  for (; t < e; ++t) acc = __fadd_rn(acc, X[(uint64_t)sorted_ids[t] * ld + col]);
  S[(uint64_t)c * ldc + col] = acc;
}

// centroid = sum / (count as f32), empty cluster -> zero vector (ivfflat.rs:58-68); counts are the GLOBAL member counts
__global__ void finish_centroids_kernel(const float* S, const uint32_t* counts, uint32_t k, uint32_t ldc, float* Cnew) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)k * ldc) return;
  const uint32_t n = counts[i / ldc];
  Cnew[i] = n ? __fdiv_rn(S[i], (float)n) : 0.0f;
}

int32_t km_update_sums(const float* X, uint32_t ld, uint32_t d, const uint32_t* sorted_ids, const uint32_t* starts, uint32_t k, float* S,
                       uint32_t ldc, hipStream_t st) {
  if (k == 0) return VERS_OK;
  hipLaunchKernelGGL(update_sums_kernel, dim3(k, (ldc + 255) / 256), dim3(256), 0, st, X, ld, d, sorted_ids, starts, S, ldc);
  VERS_HIP_TRY(hipGetLastError());
  return VERS_OK;
}

int32_t km_finish_centroids(const float* S, const uint32_t* counts, uint32_t k, uint32_t ldc, float* Cnew, hipStream_t st) {
  if (k == 0) return VERS_OK;
  hipLaunchKernelGGL(finish_centroids_kernel, dim3((unsigned)(((uint64_t)k * ldc + 255) / 256)), dim3(256), 0, st, S, counts, k, ldc, Cnew);
  VERS_HIP_TRY(hipGetLastError());
  return VERS_OK;
}

int32_t km_update(const float* X, uint32_t ld, uint32_t d, const uint32_t* sorted_ids, const uint32_t* starts, const uint32_t* counts, uint32_t k,
                  float* Cnew, uint32_t ldc, hipStream_t st) {
  if (k == 0) return VERS_OK;
  // single process: the running sums live in Cnew itself (zeroed, summed, divided in place)
  VERS_HIP_TRY(hipMemsetAsync(Cnew, 0, (size_t)k * ldc * sizeof(float), st));
  if (int32_t rc = km_update_sums(X, ld, d, sorted_ids, starts, k, Cnew, ldc, st)) return rc;
  return km_finish_centroids(Cnew, counts, k, ldc, Cnew, st);
}

// ---- cost -------------------------------------------------------------------------------
// The fold is one dependent chain of n f32 adds (the reference's .fold(0.0, |acc, val| acc + val)); what can be
// taken off the chain is everything else: one wave loads 256 values per step with a single coalesced float4 per
// lane (double buffered), v_readlane hands them to the chain in index order as scalar operands.  ~6 cycles per
// element instead of one scalar-thread load + add (~56 cycles): 0.26 s -> ~0.03 s at n = 10M.
__global__ __launch_bounds__(kWave) void cost_fold_kernel(const float* v, uint64_t n, const float* init, float* out) {
  if (blockIdx.x != 0) return;
  const int lane = threadIdx.x;
  float acc = init ? *init : 0.0f;  // the fold continues the previous rank's partial cost when the points are sharded
  const uint64_t n_blk = n / 256;
  const bool aligned = (reinterpret_cast<uintptr_t>(v) & 15u) == 0;
  uint64_t done = 0;
  if (aligned && n_blk) {
    const f32x4* v4 = reinterpret_cast<const f32x4*>(v);
    f32x4 cur = v4[lane];
    for (uint64_t bk = 0; bk < n_blk; ++bk) {
      const uint64_t nb = bk + 1 < n_blk ? bk + 1 : bk;
      const f32x4 nxt = v4[nb * 64 + lane];  // in flight under this block's chain
#pragma unroll
      for (int j = 0; j < kWave; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          acc = __fadd_rn(acc, __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(cur[u]), j)));
      cur = nxt;
    }
    done = n_blk * 256;
  }
  for (uint64_t i = done; i < n; ++i) acc = __fadd_rn(acc, v[i]);  // tail (and unaligned input): every lane, same value
  if (lane == 0) *out = acc;
}

int32_t km_cost_fold(const float* mind, uint64_t n, const float* init_dev, float* out_dev, hipStream_t st) {
  hipLaunchKernelGGL(cost_fold_kernel, dim3(1), dim3(64), 0, st, mind, n, init_dev, out_dev);
  VERS_HIP_TRY(hipGetLastError());
  return VERS_OK;
}

__global__ void differs_kernel(const uint32_t* a, const uint32_t* b, uint64_t n, uint32_t* flag) {
  bool diff = false;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    diff |= a[i] != b[i];
  if (diff) atomicOr(flag, 1u);
}

int32_t km_differs(const float* a, const float* b, uint64_t n_words, uint32_t* flag_dev, hipStream_t st) {
  VERS_HIP_TRY(hipMemsetAsync(flag_dev, 0, sizeof(uint32_t), st));
  if (n_words == 0) return VERS_OK;
  uint64_t blocks = (n_words + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(differs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const uint32_t*)a, (const uint32_t*)b, n_words,
                     flag_dev);
  VERS_HIP_TRY(hipGetLastError());
  return VERS_OK;
}

// D(x_i, c_{a_i}) for a GIVEN assignment (standalone cost primitive; build_index gets these
// for free from the assign pass).  One lane per point, strictly ordered columns.
__global__ void pair_dist_kernel(const float* X, const float* C, const uint32_t* assign, uint64_t n, uint32_t ld,
                                 float* out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f32x4* x = reinterpret_cast<const f32x4*>(X + i * ld);
  const f32x4* c = reinterpret_cast<const f32x4*>(C + (uint64_t)assign[i] * ld);
  float acc = 0.0f;
  for (uint32_t j = 0; j < ld / 4; ++j) {
    const f32x4 xv = x[j], cv = c[j];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float t = __fsub_rn(xv[u], cv[u]);
      acc = __fadd_rn(acc, __fmul_rn(t, t));
    }
  }
  out[i] = acc;
}

}  // namespace vers

// ======================================================================================
// C ABI: k-means primitives on host arrays (unit-test and integration surface).
// ======================================================================================
using namespace vers;

namespace {

int32_t upload_rows(DevBuf& buf, const float* rows, uint64_t n, uint64_t stride_bytes, uint32_t d, uint32_t ld) {
  if (int32_t rc = buf.reserve((n ? n : 1) * (size_t)ld * sizeof(float))) return rc;
  if (n == 0) return VERS_OK;
  if (ld != d) VERS_HIP_TRY(hipMemset(buf.p, 0, n * (size_t)ld * sizeof(float)));
  VERS_HIP_TRY(hipMemcpy2D(buf.p, (size_t)ld * 4, rows, stride_bytes, (size_t)d * 4, n, hipMemcpyHostToDevice));
  return VERS_OK;
}

int32_t device_cus(int device, int* n_cu) {
  int cnt = 0;
  VERS_HIP_TRY(hipGetDeviceCount(&cnt));
  if (device < 0 || device >= cnt) return fail(VERS_ERR_INVALID, "no such device");
  hipDeviceProp_t prop;
  VERS_HIP_TRY(hipGetDeviceProperties(&prop, device));
  *n_cu = prop.multiProcessorCount;
  return VERS_OK;
}

int32_t check_status_word(DevBuf& status, uint32_t k) {
  uint32_t st = 0;
  VERS_HIP_TRY(hipMemcpy(&st, status.p, sizeof(st), hipMemcpyDeviceToHost));
  VERS_HIP_TRY(hipMemset(status.p, 0, sizeof(st)));
  // the reference only panics once a NaN is COMPARED, which needs at least two centroids
  if ((st & 1u) && k >= 2) return fail(VERS_ERR_NAN, "NaN distance in assign_to_clusters (reference panics)");
  return VERS_OK;
}

}  // namespace

extern "C" {

int32_t vers_kmeans_assign(int32_t device, const float* rows, uint64_t n, uint64_t row_stride_bytes, const float* centroids,
                           uint64_t k, uint64_t c_stride_bytes, uint32_t d, uint64_t* out_assign, float* out_min_dist) {
  if (d == 0 || (n && (!rows || !out_assign)) || (k && !centroids) || row_stride_bytes < (uint64_t)d * 4 ||
      (k && c_stride_bytes < (uint64_t)d * 4) || n > 0xFFFFFFFFull || k > 0xFFFFFFFFull)
    return fail(VERS_ERR_INVALID, "vers_kmeans_assign: bad arguments");
  if (n == 0) return VERS_OK;
  if (k == 0) return fail(VERS_ERR_EMPTY, "min_by over zero centroids (reference: unwrap on None)");
  int n_cu = 0;
  if (int32_t rc = device_cus(device, &n_cu)) return rc;
  DeviceGuard g(device);
  const uint32_t ld = round_up(d, 4);
  KMeansScratch ws;
  DevBuf X, C, A, M;
  if (int32_t rc = upload_rows(X, rows, n, row_stride_bytes, d, ld)) return rc;
  if (int32_t rc = upload_rows(C, centroids, k, c_stride_bytes, d, ld)) return rc;
  if (int32_t rc = A.reserve(n * sizeof(uint32_t))) return rc;
  if (int32_t rc = M.reserve(n * sizeof(float))) return rc;
  if (int32_t rc = ws.status.reserve(16)) return rc;
  VERS_HIP_TRY(hipMemset(ws.status.p, 0, 16));
  if (int32_t rc = (km_use_mfma(n, (uint32_t)k, d) ? km_assign_mfma : km_assign)(X.as<float>(), ld, n, C.as<float>(), ld, (uint32_t)k, d,
                                                                                 A.as<uint32_t>(), M.as<float>(), ws, n_cu, nullptr, 0))
    return rc;
  VERS_HIP_TRY(hipDeviceSynchronize());
  if (int32_t rc = check_status_word(ws.status, (uint32_t)k)) return rc;
  std::vector<uint32_t> a32(n);
  VERS_HIP_TRY(hipMemcpy(a32.data(), A.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  for (uint64_t i = 0; i < n; ++i) out_assign[i] = a32[i];
  if (out_min_dist) VERS_HIP_TRY(hipMemcpy(out_min_dist, M.p, n * sizeof(float), hipMemcpyDeviceToHost));
  return VERS_OK;
}

// assign_to_clusters (ivfflat.rs:29-46) on device-resident rows and centroids: what a host that streams a corpus larger than
// one GPU through a trained quantiser calls per chunk.
static __global__ void widen_assign_kernel(const uint32_t* in, uint64_t n, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
int32_t vers_kmeans_assign_dev(int32_t device, const float* rows_dev, uint64_t n, uint64_t ld_floats, const float* centroids_dev, uint64_t k,
                               uint64_t c_ld_floats, uint32_t d, uint64_t* out_assign_dev, float* out_min_dist_dev) {
  if (d == 0 || (n && (!rows_dev || !out_assign_dev)) || (k && !centroids_dev) || ld_floats < d || ld_floats % 4 || ld_floats > 0x3FFFFFFFull ||
      (k && c_ld_floats < d) || n > 0xFFFFFFFFull || k > 0xFFFFFFFFull)
    return fail(VERS_ERR_INVALID, "vers_kmeans_assign_dev: bad arguments (ld_floats must be >= d and a multiple of 4)");
  // The scratch of the matrix-core path (centroid operands, up to ~1 GiB of query blocks and GEMM workspaces at k = 65536) is kept PER DEVICE
  // between calls -- a streamed corpus is hundreds of calls of one shape -- under a mutex: calls on one device take turns (they use the null
  // stream and end with a device synchronisation anyway).  A call with n == 0 RELEASES the device's scratch (rounds 4-5 kept one per calling
  // THREAD and never freed it: a host streaming from a pool of worker threads leaked that much HBM per thread, inside vers_mem_stats' figure).
  struct Scratch { KMeansScratch ws; DevBuf C, A, M; };
  static std::mutex mu;
  static std::map<int, Scratch*> per_device;  // (never destroyed at exit: DevBuf's destructor must not run after the runtime is gone)
  std::lock_guard<std::mutex> lk(mu);
  if (n == 0) {
    auto it = per_device.find(device);
    if (it != per_device.end()) {
      DeviceGuard g0(device);
      delete it->second;
      per_device.erase(it);
    }
    return VERS_OK;
  }
  if (k == 0) return fail(VERS_ERR_EMPTY, "min_by over zero centroids (reference: unwrap on None)");
  int n_cu = 0;
  if (int32_t rc = device_cus(device, &n_cu)) return rc;
  DeviceGuard g(device);
  const uint32_t ld = round_up(d, 4);
  Scratch*& s = per_device[device];
  if (s == nullptr) s = new Scratch();
  const size_t cbytes = (size_t)k * ld * sizeof(float);
  if (int32_t rc = s->C.reserve(cbytes)) return rc;
  if (int32_t rc = s->A.reserve(n * sizeof(uint32_t))) return rc;
  if (int32_t rc = s->M.reserve(n * sizeof(float))) return rc;
  if (int32_t rc = s->ws.status.reserve(16)) return rc;
  VERS_HIP_TRY(hipMemset(s->ws.status.p, 0, 16));
  if (ld != d) VERS_HIP_TRY(hipMemset(s->C.p, 0, cbytes));  // (the kernels read the centroids' padding columns: zeros)
  VERS_HIP_TRY(hipMemcpy2D(s->C.p, (size_t)ld * 4, centroids_dev, (size_t)c_ld_floats * 4, (size_t)d * 4, k, hipMemcpyDeviceToDevice));
  if (int32_t rc = (km_use_mfma(n, (uint32_t)k, d) ? km_assign_mfma : km_assign)(rows_dev, (uint32_t)ld_floats, n, s->C.as<float>(), ld, (uint32_t)k, d,
                                                                                 s->A.as<uint32_t>(), s->M.as<float>(), s->ws, n_cu, nullptr, 0))
    return rc;
  hipLaunchKernelGGL(widen_assign_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, s->A.as<uint32_t>(), n, out_assign_dev);
  VERS_HIP_TRY(hipGetLastError());
  if (out_min_dist_dev) VERS_HIP_TRY(hipMemcpyAsync(out_min_dist_dev, s->M.p, n * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
  VERS_HIP_TRY(hipDeviceSynchronize());
  return check_status_word(s->ws.status, (uint32_t)k);
}

int32_t vers_build_stats(double* out, int32_t reset) {
  km_timers_collect();
  std::lock_guard<std::mutex> lk(g_bs_mu);
  if (out) {
    out[0] = g_bs.gemm_ms; out[1] = g_bs.gemm_launches; out[2] = g_bs.gemm_flop; out[3] = g_bs.assign_ms; out[4] = g_bs.assign_passes;
    out[5] = g_bs.update_ms; out[6] = g_bs.cost_ms; out[7] = g_bs.redone_points;
  }
  if (reset) g_bs = BuildStats{};
  return VERS_OK;
}

int32_t vers_build_phases(double* out, int32_t reset) {
  km_timers_collect();
  std::lock_guard<std::mutex> lk(g_bs_mu);
  if (out) {
    out[0] = g_bs.total_ms; out[1] = g_bs.alloc_ms; out[2] = g_bs.assign_ms; out[3] = g_bs.assign_first_ms; out[4] = g_bs.assign_passes;
    out[5] = g_bs.update_ms; out[6] = g_bs.cost_ms; out[7] = g_bs.install_ms; out[8] = g_bs.derive_ms;
    out[9] = g_bs.total_ms - (g_bs.alloc_ms + g_bs.assign_ms + g_bs.update_ms + g_bs.cost_ms + g_bs.install_ms + g_bs.derive_ms);
  }
  if (reset) g_bs = BuildStats{};
  return VERS_OK;
}

int32_t vers_assign_stats(uint64_t* out_points, uint64_t* out_fallbacks, int32_t reset) {
  if (out_points) *out_points = g_mfma_points.load();
  if (out_fallbacks) *out_fallbacks = g_mfma_fallbacks.load();
  if (reset) { g_mfma_points = 0; g_mfma_fallbacks = 0; }
  return VERS_OK;
}

int32_t vers_kmeans_update(int32_t device, const float* rows, uint64_t n, uint64_t row_stride_bytes, const uint64_t* assign,
                           uint64_t k, uint32_t d, float* out_centroids) {
  if (d == 0 || (n && (!rows || !assign)) || (k && !out_centroids) || row_stride_bytes < (uint64_t)d * 4 ||
      n > 0xFFFFFFFFull || k > 0xFFFFFFFFull)
    return fail(VERS_ERR_INVALID, "vers_kmeans_update: bad arguments");
  if (k == 0) return VERS_OK;
  for (uint64_t i = 0; i < n; ++i)
    if (assign[i] >= k) return fail(VERS_ERR_INVALID, "vers_kmeans_update: assignment out of range");
  int n_cu = 0;
  if (int32_t rc = device_cus(device, &n_cu)) return rc;
  DeviceGuard g(device);
  const uint32_t ld = round_up(d, 4);
  KMeansScratch ws;
  DevBuf X, A, S, CN;
  if (int32_t rc = upload_rows(X, rows, n, row_stride_bytes, d, ld)) return rc;
  std::vector<uint32_t> a32(n ? n : 1);
  for (uint64_t i = 0; i < n; ++i) a32[i] = (uint32_t)assign[i];
  if (int32_t rc = A.reserve((n ? n : 1) * sizeof(uint32_t))) return rc;
  if (n) VERS_HIP_TRY(hipMemcpy(A.p, a32.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
  if (int32_t rc = S.reserve((n ? n : 1) * sizeof(uint32_t))) return rc;
  if (int32_t rc = ws.counts.reserve((2 * k + 1) * sizeof(uint32_t))) return rc;
  if (int32_t rc = CN.reserve(k * (size_t)ld * sizeof(float))) return rc;
  uint32_t* counts = ws.counts.as<uint32_t>();
  uint32_t* starts = counts + k;
  if (int32_t rc = km_group(A.as<uint32_t>(), (uint32_t)n, (uint32_t)k, S.as<uint32_t>(), counts, starts, ws, nullptr)) return rc;
  if (int32_t rc = km_update(X.as<float>(), ld, d, S.as<uint32_t>(), starts, counts, (uint32_t)k, CN.as<float>(), ld, nullptr)) return rc;
  VERS_HIP_TRY(hipMemcpy2D(out_centroids, (size_t)d * 4, CN.p, (size_t)ld * 4, (size_t)d * 4, k, hipMemcpyDeviceToHost));
  return VERS_OK;
}

int32_t vers_kmeans_cost(int32_t device, const float* rows, uint64_t n, uint64_t row_stride_bytes, const float* centroids,
                         uint64_t k, uint64_t c_stride_bytes, const uint64_t* assign, uint32_t d, float* out_cost) {
  if (d == 0 || !out_cost || (n && (!rows || !assign || !centroids)) || row_stride_bytes < (uint64_t)d * 4 ||
      (k && c_stride_bytes < (uint64_t)d * 4) || n > 0xFFFFFFFFull)
    return fail(VERS_ERR_INVALID, "vers_kmeans_cost: bad arguments");
  for (uint64_t i = 0; i < n; ++i)
    if (assign[i] >= k) return fail(VERS_ERR_INVALID, "vers_kmeans_cost: assignment out of range");
  int n_cu = 0;
  if (int32_t rc = device_cus(device, &n_cu)) return rc;
  DeviceGuard g(device);
  const uint32_t ld = round_up(d, 4);
  DevBuf X, C, A, M, O;
  if (int32_t rc = upload_rows(X, rows, n, row_stride_bytes, d, ld)) return rc;
  if (int32_t rc = upload_rows(C, centroids, k, c_stride_bytes, d, ld)) return rc;
  std::vector<uint32_t> a32(n ? n : 1);
  for (uint64_t i = 0; i < n; ++i) a32[i] = (uint32_t)assign[i];
  if (int32_t rc = A.reserve((n ? n : 1) * sizeof(uint32_t))) return rc;
  if (n) VERS_HIP_TRY(hipMemcpy(A.p, a32.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
  if (int32_t rc = M.reserve((n ? n : 1) * sizeof(float))) return rc;
  if (int32_t rc = O.reserve(16)) return rc;
  if (n) {
    hipLaunchKernelGGL(pair_dist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, X.as<float>(), C.as<float>(),
                       A.as<uint32_t>(), n, ld, M.as<float>());
    VERS_HIP_TRY(hipGetLastError());
  }
  if (int32_t rc = km_cost_fold(M.as<float>(), n, nullptr, O.as<float>(), nullptr)) return rc;
  VERS_HIP_TRY(hipMemcpy(out_cost, O.p, sizeof(float), hipMemcpyDeviceToHost));
  return VERS_OK;
}

}  // extern "C"
