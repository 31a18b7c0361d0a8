// ivf_handle.hpp -- the IVFFlat handle behind vers_ivf_* (ivfflat.rs:8-15: a device cache of the reference's five fields), the
// per-call workspaces and what the translation units of the index share.  Round 4 split the former single 3,400-line ivf.hip:
//   ivf_handle.hip  lifecycle, workspace pool, status words, options, getters
//   ivf_build.hip   build_index / upload / add (ivfflat.rs:18-136, 200-213): storage layout, row placement, k-means driver, sharded build
//   ivf_plan.hip    coarse quantiser + planning of a search (ivfflat.rs:155-161 and the walk's plan, 166-195)
//   ivf_search.hip  list scans, merges, exact finish and the search entry points (ivfflat.rs:153-198; utils.rs:68-82 on the stored rows)
//   ivf_hooks.hip   measurement and test hooks
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <vector>

#include "kmeans.hpp"
#include "plan.hip.h"
#include "util.hip.h"
#include "wide.hip.h"

namespace vers {
void shard_plan(const uint64_t* lens, uint64_t k, uint32_t world, uint8_t* owner);
constexpr int32_t kRetrySpill = 1001;  // internal: reference-mode spill ran past the ranked lists, retry deeper if possible
}  // namespace vers

using namespace vers;

// =================================================================================================
// Everything a search call MUTATES lives in a workspace, not in the handle: scratch buffers, status words, the timing
// ring, the host-pointer staging, the look-ahead slots.  Search_approximate(&self) is legal from many threads in the
// reference (plain Vecs behind a shared borrow); here every call leases a workspace from the handle's pool
// (a lease = a pop under a short mutex), so concurrent callers enqueue side by side on their own streams instead of
// queueing behind one per-handle mutex as in round 1.  A workspace that changes streams is ordered by its `done` event.
struct SearchWs {
  DevBuf gbuf;         // G [M_pad][k_pad] of the batched coarse quantiser
  bool ref_deep = false;     // reference-mode retry: rank 64 lists with the exact coarse quantiser (no slack needed)
  bool ref_all = false;      // last resort of a reference-mode host call: every list is ranked (the spill may walk through all of them)
  bool ref_shallow = false;  // host-pointer calls try 16 ranked lists first (a spill past the nearest few lists is rare)
  DevBuf seg_bounds, stamps, quad_counter, fb_part, fb_ctr, c1_ctr;  // (c1_ctr: coarse1_kernel's finished-blocks counter)
  DevBuf clower, lower;  // lower bounds of multi-pass results (coarse ranking of more than 64 lists; top_k > 64)
  DevBuf qp, qil, cpart, probe, pj, lists, pairs, items, groups, qblocks, partials, status, o_ids, o_dist, o_cnt, xpart;
  DevBuf g_send, g_recv;  // sharded search without the host in the loop: this rank's [2][b][top_k] partial | the world's
  bool g_poisoned = false;  // g_send holds the poison mark of a locally failed batch (cleared before the next partial is written)
  static constexpr uint32_t kEvRing = 64;  // scan-launch timing ring (measurement hook)
  hipEvent_t ev0[kEvRing] = {}, ev1[kEvRing] = {};
  hipEvent_t evc[3] = {};  // batched coarse quantiser of the most recent search: before the GEMM | after it | after select / re-score
  bool evc_valid = false;
  hipEvent_t evf = nullptr;  // recorded behind the exact finish of the most recent matrix-core batch (its start = the scan's ev1)
  bool evf_valid = false;
  uint32_t evf_slot = 0;
  uint64_t ev_count = 0;
  bool ev_on = true;  // this search brackets its list-scan launch with event records (scan_events_ref)
  size_t ivf_bounds_off = 0;  // pruning bounds live behind the partial slots (one memset)
  // host-pointer entry points: one pinned staging buffer, one device buffer for queries, one for the packed
  // results, one stream -- a call is one H2D copy, the kernels, one D2H copy and ONE synchronisation
  DevBuf io_q, io_out;
  void* io_pin = nullptr;
  size_t io_pin_cap = 0;
  uint32_t* st_host = nullptr;  // set by a host-pointer single-query call: the last merge launch stores the status word there (pinned)
  hipStream_t io_stream = nullptr;
  // Coarse quantiser one batch ahead (vers_ivf_coarse_ahead_dev): staged queries + ranked lists of the NEXT batch are
  // computed on a side stream; two slots alternate (one is read by the search in flight while the other is written).
  // ready: recorded on the side stream after the slot's kernels; freed: recorded on the consuming search's stream after
  // its last kernel.  (Per workspace: a single-threaded serving loop always leases the same one.)
  struct CoarseAhead {
    DevBuf qp, probe;
    const float* q_dev = nullptr;
    uint64_t ldq_in = 0;
    uint32_t b = 0, P = 0;
    bool valid = false, ready_rec = false, freed_rec = false;
    hipEvent_t ready = nullptr, freed = nullptr;
  };
  CoarseAhead ahead[2];
  uint32_t ahead_next = 0;
  hipStream_t ahead_stream = nullptr;
  hipEvent_t ahead_in = nullptr;
  // status words: [0] latched by _dev calls and reported by vers_ivf_poll; [1] used by host-pointer calls and add, which
  // synchronise and consume it themselves -- so neither side eats the other's bits
  // Word 0 of a DEVICE-pointer call is not the workspace's but its STREAM's (vers_ivf::stream_word): vers_ivf_poll(stream)
  // then reports exactly the calls that ran on that stream -- whichever workspaces they leased, whatever other threads run.
  uint32_t st_slot = 0;
  uint32_t* st_dev = nullptr;  // the leasing _dev call's stream word
  uint32_t* st_word() const { return st_slot == 0 && st_dev ? st_dev : status.as<uint32_t>() + st_slot; }
  // geometry of the most recent matrix-core list scan on this workspace (TEST HOOK vers_ivf_test_last_vals)
  struct LastPre { bool valid = false; uint32_t b = 0, P = 0, S_max = 0, kp = 0, top_k = 0; const float* qp = nullptr; int shadow = 0; } last_pre;
  GroupTotals last_tot{};
  const GroupTotals* tot_dev = nullptr;  // device totals of the last planned search
  bool tot_valid = false;
  // lease bookkeeping
  hipEvent_t done = nullptr;         // recorded on the leasing call's stream when the call has queued its last operation
  hipStream_t last_stream = nullptr;
  bool used = false;
};
extern thread_local SearchWs* W;  // the workspace leased by the call running on this thread (ivf_handle.hip)

struct vers_ivf {
  int device = 0, n_cu = 256;
  uint32_t d = 0;
  int metric = 0;    // VERS_METRIC_L2SQ (the reference) or VERS_METRIC_COSDIST in every distance of build / add / search
  uint32_t ldx = 0;  // pitch of row-major matrices (X, centroids): round_up(d, 4)
  uint32_t ld = 0;   // columns of blocked matrices and padded queries: round_up(d, kColAlign)
  uint32_t ldq = 0;  // == ld
  // index state (device cache of the reference's five fields, ivfflat.rs:9-15): read-only for searches
  uint32_t k = 0;         // num_centroids; 0 = no index / nothing kept
  uint64_t n_total = 0;   // assignments.len(): next vec_id handed out by add
  DevBuf centroids;    // [k][ldx] row-major (k-means, read-back)
  DevBuf centroids_b;  // the same in lane-transposed tiles (exact coarse quantiser)
  // MFMA pre-selection of the batched coarse quantiser (gemm.hip.h)
  DevBuf centroids_g;  // row-major [k_pad][ldq], zero padded
  DevBuf centroids_gs; // the same split into bf16 hi | lo halves [2][k_pad][ldq]
  DevBuf cnorm;        // |c|^2 [k_pad], +inf in the padding
  DevBuf coarse_stat;  // u32: queries that failed the certificate and were re-done exactly
  float cmax2 = 0.0f;
  uint32_t k_pad = 0;
  std::atomic<uint64_t> mfma_batches{0};
  DevBuf rows, row_ids, list_off, list_len;
  DevBuf tile_list;  // [cap_rows / 64] the list a storage tile belongs to (build: gather_tiles_kernel)
  // SLOT space: the per-batch planning tables, the work items and the scan kernels address a list by its SLOT =
  // rank among the lists by descending length (ties by index).  plan_query translates a centroid index into a slot
  // once (list_slot); every table the later stages read is then contiguous in work order: the group step's prefix
  // sums run longest list first -- the dynamic hand-out ends on short quads instead of starting a 5x longer list on
  // the last free CU (8-way sharded list scan 744 -> 670 us, same box) -- without a single gather.
  DevBuf list_slot;        // [k] centroid index -> slot
  DevBuf slot_off, slot_len;  // list_off / list_len in slot order
  std::vector<uint32_t> h_slot;
  // Reference mode from device pointers cannot come back for a deeper ranking (the host-pointer entry retries; a _dev call
  // only latches a status), so the depth is decided UP FRONT from what the host knows: the walk of ivfflat.rs:166-195 stops
  // once top_k rows are gathered, and ANY P lists hold at least the sum of the P SHORTEST lists' lengths.  len_asc_prefix[i]
  // = rows in the i + 1 shortest lists (as of build / upload; add() only lengthens lists, the bound stays valid).
  std::vector<uint64_t> len_asc_prefix;
  uint32_t lists_that_always_suffice(uint32_t top_k) const {  // smallest P such that every set of P lists holds >= top_k rows (k if none)
    const auto it = std::lower_bound(len_asc_prefix.begin(), len_asc_prefix.end(), (uint64_t)top_k);
    return it == len_asc_prefix.end() ? (uint32_t)len_asc_prefix.size() : (uint32_t)(it - len_asc_prefix.begin()) + 1u;
  }
  std::vector<uint32_t> h_off, h_len, h_cap;  // h_len = GLOBAL list lengths; h_off/h_cap only meaningful for owned lists
  // sharding by cluster across GPUs (one process per GPU): this handle stores only lists with owner == rank
  uint32_t rank = 0, world = 1;
  std::vector<uint8_t> h_owner;
  DevBuf owner;
  uint64_t cap_rows = 0;
  uint32_t max_len = 0;
  KMeansScratch km;  // build scratch (build / upload hold the handle exclusively)
  // A streamed upload in progress (vers_ivf_upload_begin .. _end, ivf_build.hip): the storage plan is made from the GLOBAL
  // list lengths at begin, chunks of (rows, assignments) fill the owned lists in ascending vec id, nothing of size n_total
  // is ever held.  h->k stays 0 until _end: searches in between see an empty index.
  struct UploadState {
    bool open = false;
    uint32_t k = 0;
    uint64_t n_total = 0, seen = 0, cap_rows = 0;
    DevBuf fill;                   // u32 [k] rows placed in every OWNED list so far | u32 [k] rows of EVERY list counted on the device
    std::vector<uint32_t> h_seen;  // rows of every list counted by the host path (it sends only the owned ones)
    DevBuf a32, sorted, ids, a64, stage, bad;  // per-chunk scratch: assignments as u32, cluster-sorted order, vec ids, staging
    void* pin = nullptr;           // pinned staging of the host path: TWO halves of (rows | vec ids | assignments of one sub-chunk)
    size_t pin_cap = 0;
    hipEvent_t half_free[2] = {nullptr, nullptr};  // recorded behind the copies + placement that read a half
    void close() {
      open = false;
      for (auto& e : half_free)
        if (e) { (void)hipEventDestroy(e); e = nullptr; }
      fill.release(); a32.release(); sorted.release(); ids.release(); a64.release(); stage.release(); bad.release();
      h_seen.clear(); h_seen.shrink_to_fit();
      if (pin) { (void)hipHostFree(pin); pin = nullptr; pin_cap = 0; }
    }
  } up;
  // matrix-core list scan (prescan.hip.h): |x|^2 per storage row, [0] max |x|^2 bits, [1] certificate failures (running)
  DevBuf xnorm, pre_misc;
  // fp16 shadow of the rows for the matrix-core pre-selection (+50 % corpus memory; VERS_SHADOW=0 or a failed
  // allocation: the f32 rows feed it).  The wider certificate window makes it sensitive to data with many near-ties:
  // the failure counter is watched through a pinned word and the shadow is switched off for the handle when more
  // than 1/8 of the queries had to be re-scanned exactly.
  DevBuf rows_bf;
  // Row-major second copy of the stored rows for the exact finish: a candidate row of the lane-transposed tile layout is
  // 192 separate 16-byte pieces (one per 64-byte sector: 4x the useful bytes); row-major it is 3 KB of whole sectors.
  // ON whenever the rows take at most a quarter of the device's memory (VERS_ROWMAJOR=-1, the default since round 4; 1 =
  // always, 0 = never): measured at cfg3 over every rank of 1 / 8 ranks, same box -- the step with three batches in flight
  // 2.456 -> 2.379 ms on one GPU and 0.461 -> 0.437 ms at 8 ranks (exact finish 80 -> ~55 us).  It is optional memory: +100 %
  // of the f32 rows (81 GB instead of 49 GB of 288 at N = 10M d = 768); a failed allocation or rows beyond the quarter
  // (N > 23M at d = 768) leave the tile gather in charge.  Same bits either way (tests run both).
  DevBuf rows_rm;
  std::atomic<bool> shadow_off{false};
  bool shadow_valid = false;            // rows_bf mirrors every stored row of the CURRENT index (written under the exclusive lock)
  uint32_t* fail_watch = nullptr;       // pinned: cumulative certificate failures as of the last finished batch
  std::atomic<uint64_t> shadow_queries{0};  // queries sent through the shadow path since the counter was last zeroed
  std::atomic<uint64_t> pre_batches{0};
  std::atomic<uint64_t> ahead_used{0};  // searches that consumed a look-ahead slot (statistics)
  // A look-ahead request is DEFERRED: vers_ivf_coarse_ahead_dev only notes it, and the next search on the handle starts
  // it right behind its own list-scan launch -- the side stream then works under that search's exact finish (a chain of
  // dependent row gathers: 9 % VALU-active, 82 % of its wave cycles waiting) instead of competing with the scan, which
  // fills every CU and the HBM pipe (round 1 started it at once and measured no gain).
  struct PendingAhead {
    bool set = false;
    const float* q_dev = nullptr;
    uint64_t ldq_in = 0;
    uint32_t b = 0, nprobe = 0;
  } pending;  // (guarded by pool_mu)
  // Status words of the device-pointer calls, one per stream the handle has seen (first come, first served; streams beyond
  // the table share its last word): latched by the kernels of the calls queued on that stream, read and cleared ON that
  // stream by vers_ivf_poll -- so a poll never consumes another stream's panic, and never clears a word while a kernel of
  // its own stream can still set it.
  // (Round 3 kept 64 words and let the streams beyond share the last one: a poll on one of them could consume another stream's
  // status.  Now 1024 words and an error beyond -- no word is ever shared.)
  static constexpr uint32_t kStreamWords = 1024;
  DevBuf st_words;
  std::mutex st_mu;
  std::vector<hipStream_t> st_streams;
  uint32_t* st_pin = nullptr;  // pinned landing words of the polls, one per stream word (no handle-wide lock is held while a poll waits for its stream)
  // Two threads polling the SAME stream share its landing word: copy / clear / wait / read is one critical section per word
  // (interleaved, the second poll's copy could land a 0 over the first one's latched status before it is read).
  static constexpr uint32_t kPollLocks = 64;
  std::mutex st_poll_mu[kPollLocks];
  int32_t stream_word(hipStream_t st, uint32_t** out, uint32_t** out_pin = nullptr, std::mutex** out_mu = nullptr) {
    std::lock_guard<std::mutex> lk(st_mu);
    if (!st_words.p) {
      if (int32_t rc = st_words.reserve(kStreamWords * sizeof(uint32_t))) return rc;
      VERS_HIP_TRY(hipMemset(st_words.p, 0, kStreamWords * sizeof(uint32_t)));
      VERS_HIP_TRY(hipHostMalloc((void**)&st_pin, kStreamWords * sizeof(uint32_t), hipHostMallocDefault));
    }
    uint32_t i = 0;
    while (i < st_streams.size() && st_streams[i] != st) ++i;
    if (i == st_streams.size()) {
      if (i >= kStreamWords) return fail(VERS_ERR_INVALID, "more than 1024 distinct streams used with one handle: no status word left for this one");
      st_streams.push_back(st);
    }
    *out = st_words.as<uint32_t>() + i;
    if (out_pin) *out_pin = st_pin + i;
    if (out_mu) *out_mu = &st_poll_mu[i % kPollLocks];
    return VERS_OK;
  }
  // searches / reads hold `index` shared, build / upload / add / set_* exclusively
  std::shared_mutex index;
  std::mutex pool_mu;
  std::condition_variable pool_cv;
  std::vector<std::unique_ptr<SearchWs>> pool;
  std::vector<SearchWs*> free_ws;
  SearchWs* last_ws = nullptr;  // the measurement hooks (vers_ivf_last_scan, ...) read the workspace of the most recent call
  static constexpr size_t kMaxWs = 16;
};

namespace vers {
namespace ivf {

// ---- ivf_handle.hip ----------------------------------------------------------------------------------------------
int32_t ws_init(SearchWs& w);
void ws_destroy(SearchWs& w);
int32_t sync_status(vers_ivf* h, hipStream_t st);            // the word of the _dev calls queued on `st`, read and cleared on `st`
int32_t status_to_rc(vers_ivf* h, uint32_t s, uint32_t slot);  // maps (and clears) the device status word of a finished search

// One call's lease of a workspace (see SearchWs).  order_on(stream): the call is about to queue work on `stream`; if
// the workspace was last used on another stream, that stream's work on it must finish first.
struct WsLease {
  vers_ivf* h;
  SearchWs* ws = nullptr;
  SearchWs* prev;
  hipStream_t st = nullptr;
  int32_t rc = VERS_OK;
  // dev_stream: a device-pointer call names the stream it will queue on.  It gets the free workspace that last ran on
  // that stream if there is one (no cross-stream ordering needed: batches a host keeps in flight on two or three streams
  // each get their own scratch and overlap on the GPU -- the small latency-bound kernels of one batch under the list scan
  // of another), else a fresh one while the pool may grow, else the most recently freed (ordered by its `done` event).
  explicit WsLease(vers_ivf* hh, bool dev = false, hipStream_t dev_stream = nullptr) : h(hh), prev(W) {
    {
      std::unique_lock<std::mutex> lk(h->pool_mu);
      for (;;) {
        if (dev && !h->free_ws.empty()) {
          size_t pick = h->free_ws.size();
          for (size_t i = h->free_ws.size(); i-- > 0;)
            if (h->free_ws[i]->used && h->free_ws[i]->last_stream == dev_stream) { pick = i; break; }
          if (pick == h->free_ws.size() && h->pool.size() >= vers_ivf::kMaxWs) pick = h->free_ws.size() - 1;
          if (pick == h->free_ws.size())
            for (size_t i = h->free_ws.size(); i-- > 0;)
              if (!h->free_ws[i]->used) { pick = i; break; }
          if (pick != h->free_ws.size()) { ws = h->free_ws[pick]; h->free_ws.erase(h->free_ws.begin() + (long)pick); break; }
        } else if (!h->free_ws.empty()) { ws = h->free_ws.back(); h->free_ws.pop_back(); break; }
        if (h->pool.size() < vers_ivf::kMaxWs) {
          h->pool.emplace_back(new SearchWs());
          ws = h->pool.back().get();
          break;
        }
        h->pool_cv.wait(lk);
      }
    }
    if (!ws->done) rc = ws_init(*ws);
    W = ws;
  }
  int32_t order_on(hipStream_t stream) {
    st = stream;
    if (ws->used && ws->last_stream != stream) VERS_HIP_TRY(hipStreamWaitEvent(stream, ws->done, 0));
    return h->stream_word(stream, &ws->st_dev);
  }
  ~WsLease() {
    ws->st_dev = nullptr;
    if (ws->done) {
      (void)hipEventRecord(ws->done, st);
      ws->used = true;
      ws->last_stream = st;
    }
    W = prev;
    {
      std::lock_guard<std::mutex> lk(h->pool_mu);
      h->free_ws.push_back(ws);  // LIFO: a single-threaded loop keeps getting the same workspace (and its look-ahead slots)
      h->last_ws = ws;
    }
    h->pool_cv.notify_one();
  }
};
// the measurement hooks look at the workspace of the most recent call
struct UseLastWs {
  SearchWs* prev;
  bool ok;
  explicit UseLastWs(vers_ivf* h) : prev(W) {
    std::lock_guard<std::mutex> lk(h->pool_mu);
    ok = h->last_ws != nullptr;
    if (ok) W = h->last_ws;
  }
  ~UseLastWs() { W = prev; }
};
// host-pointer calls and add run with status word 1 while they hold the handle's mutex
struct HostStatusSlot {
  SearchWs* w;
  explicit HostStatusSlot(vers_ivf*) : w(W) { w->st_slot = 1; }
  ~HostStatusSlot() { w->st_slot = 0; }
};
// vers_set_option("scan_events", v): HIP event records around every list-scan launch (vers_ivf_last_scan / vers_ivf_scan_times).
// 1 always, 0 never, 2 (default) for batches only: the two records cost a single-query call 5.5-6 us of ~100 (same-box A/B,
// scripts/bench_host_b1.py), a batch of 1024 nothing measurable.
inline std::atomic<int>& scan_events_ref() {
  static std::atomic<int> m{(int)opt_get("scan_events", 2)};
  return m;
}
// vers_set_option("host_spin", 0 | 1): a host-pointer single-query call waits for its result by spinning on the pinned status word
// (1, default) or in hipStreamSynchronize (0: rounds 1-4; same-process A/B in scripts/bench_host_b1.py)
inline std::atomic<int>& host_spin_ref() {
  static std::atomic<int> m{1};
  return m;
}
// vers_set_option("single_shadow", 0 | 1): a single query's list scan streams the fp16 shadow with the exact finish behind it (1,
// default) or the f32 rows through the ordered chains (0: rounds 1-4; same-process A/B in bench.py)
inline std::atomic<int>& single_shadow_ref() {
  static std::atomic<int> m{opt_get("single_shadow", 1) != 0 ? 1 : 0};
  return m;
}
// TEST HOOK (vers_set_option("test_fail_sharded", n)): the next n sharded searches of this process fail LOCALLY after their
// exchange buffers are reserved -- what an out-of-memory scratch reservation on one rank looks like to its peers
inline std::atomic<int>& test_fail_sharded_ref() {
  static std::atomic<int> m{0};
  return m;
}
// vers_set_option("scan_reserve_cus", n): compute units the persistent matrix-core list scan leaves free.
// -1 (default) = AUTO: kScanReserveAuto while ANOTHER batch of this handle is in flight on another stream (scan_reserve below), none
// otherwise.  A scan block holds 448 of a SIMD's 512 registers: nothing with a large footprint -- RCCL's all-gather kernel (256
// VGPRs per wave, profiles/r05_rccl_kernel_meta.txt), the other batches' coarse contraction (232) / selection (224) / grouping
// (1024-thread blocks) -- runs beside one, so with batches in flight those waited for a whole scan to leave the chip (a step was
// scan + 62 us of them).  The scan is HBM-bound: on 192 of 256 CUs it streams as fast (one batch alone: +6 %
// at 8 ranks, which is why a lone batch keeps every CU), and the reserved CUs run the other batches' latency-bound kernels and the
// exchange WHILE it streams: 8-rank step with three in flight and an RCCL-footprint exchange kernel 0.402-0.445 -> 0.370 ms, 4
// ranks 0.682 -> 0.643, one GPU 2.314 -> 2.276 (same box, scripts/emulate_shard.py RESERVE=...).
constexpr int kScanReserveAuto = 64;
inline std::atomic<int>& scan_reserve_cus_ref() {
  static std::atomic<int> m{(int)opt_get("scan_reserve_cus", -1)};
  return m;
}
// Is another batch of this handle in flight on ANOTHER stream right now (a workspace leased by another thread, or one whose last
// call -- on another stream -- has not finished)?  Decides the AUTO value of "scan_reserve_cus".
inline bool other_batches_in_flight(vers_ivf* h, const SearchWs* me, hipStream_t st) {
  std::lock_guard<std::mutex> lk(h->pool_mu);
  if (h->pool.size() - h->free_ws.size() > 1) return true;
  for (const SearchWs* w : h->free_ws)
    if (w != me && w->used && w->last_stream != st && w->done != nullptr && hipEventQuery(w->done) == hipErrorNotReady) return true;
  return false;
}
inline uint32_t scan_reserve(vers_ivf* h, hipStream_t st) {
  const int v = scan_reserve_cus_ref().load(std::memory_order_relaxed);
  const int r = v >= 0 ? v : (other_batches_in_flight(h, W, st) ? kScanReserveAuto : 0);
  return (uint32_t)std::min<int>(r, h->n_cu - 1);
}
inline std::atomic<int>& shadow_mode_ref() {  // vers_set_option("shadow", v) / VERS_SHADOW (default 1)
  static std::atomic<int> m{opt_get("shadow", 1) != 0 ? 1 : 0};
  return m;
}
inline int shadow_mode() { return shadow_mode_ref().load(std::memory_order_relaxed); }
// vers_set_option("pre_min_batch", v) (default 4): the smallest batch whose list scan runs on the matrix
// cores when its lists are shared by fewer than two queries on average (plan_search)
inline std::atomic<uint32_t>& pre_min_batch_ref() {
  static std::atomic<uint32_t> m{(uint32_t)opt_get("pre_min_batch", 4)};
  return m;
}

// Tuning / A-B knobs of the search path (options "pre_slack", "seg_rows", "prescan", "pre_narrow": core.hip's table).
struct SearchKnobs {
  int pre_slack = 0;     // "pre_slack": slack keys of the matrix-core lists
  long seg_rows = 0;     // "seg_rows"
  int pre_mode = 1;      // "prescan": 0 ordered chains for batches too, 2 every certificate fails
  bool pre_narrow = false;    // "pre_narrow": 16-query blocks in the matrix-core list scan whatever d is
};
inline SearchKnobs knobs() {
  SearchKnobs s;
  s.pre_slack = (int)opt_get("pre_slack", 0);
  s.seg_rows = (long)opt_get("seg_rows", 0);
  s.pre_mode = (int)opt_get("prescan", 1);
  s.pre_narrow = opt_get("pre_narrow", 0) != 0;
  return s;
}

inline int coarse_mode() {  // option "coarse": 1 = always exact, 2 = every certificate fails
  return (int)opt_get("coarse", 0);
}
inline bool coarse_on_matrix_cores(const vers_ivf* h, uint32_t b) { return b >= 32 && coarse_mode() != 1 && !W->ref_deep; }
// the whole condition under which coarse() ranks on the matrix cores (the contraction reads whole 128-row tiles: the staged block
// is padded to them, a caller's block used in place is a whole number of them; the selection keeps P + 16 keys: one per lane)
inline bool coarse_uses_mfma(const vers_ivf* h, const float* qp, uint32_t b, uint32_t P) {
  // (P + 16 <= 64: a key per lane in coarse_select_rescore_kernel; up to P + 32 <= kWideMaxKp: coarse_select_wide_kernel's four per lane)
  return coarse_on_matrix_cores(h, b) && (qp == W->qp.as<float>() || b % 128u == 0) && P + 32u <= kWideMaxKp;
}

// ---- ivf_plan.hip: coarse quantiser + planning -------------------------------------------------------------------------
int32_t stage_plain_queries(vers_ivf* h, const float* q_dev, uint64_t ldq_in, uint32_t b, const float** q_out, hipStream_t st);
int32_t coarse_mfma(vers_ivf* h, const float* qp, uint32_t b, uint32_t P, uint64_t* probe_out, hipStream_t st, const PlanQ* plan = nullptr);
int32_t coarse(vers_ivf* h, const float* qp, uint32_t b, uint32_t P, hipStream_t st, uint32_t* out_n_segs = nullptr,
               const PlanQ* plan = nullptr, bool* planned = nullptr);
int32_t coarse_ahead_locked(vers_ivf* h, const float* q_dev, uint64_t ldq_in, uint32_t b, uint32_t nprobe, hipStream_t st);
// starts a noted look-ahead behind whatever the caller has just queued on `st` (see vers_ivf::PendingAhead)
inline int32_t start_pending_ahead(vers_ivf* h, hipStream_t st) {
  vers_ivf::PendingAhead p;
  {
    std::lock_guard<std::mutex> lk(h->pool_mu);
    p = h->pending;
    h->pending.set = false;
  }
  if (!p.set) return VERS_OK;
  return coarse_ahead_locked(h, p.q_dev, p.ldq_in, p.b, p.nprobe, st);
}

// Everything the planning stage of a search decides and leaves behind for the scans (plan_search, ivf_plan.hip): which list
// scan runs and with what geometry, and where the per-batch tables of the leased workspace are.
struct SearchPlan {
  uint32_t P = 0;             // ranked lists per query
  int ref_mode = 0;           // nprobe == 0: the reference's own walk (nearest list + spill)
  bool one1 = false;          // single query, P <= 64: items are records (scan1_kernel)
  bool one1_pre = false;      // ... scanned on the fp16 shadow with the exact finish (scan1h_kernel; use_pre is set too)
  int QG = 1;                 // queries per group of the list scan (kPreQ with use_pre)
  bool use_pre = false;       // matrix-core list scan + exact finish (prescan.hip.h)
  bool use_shadow = false;    // ... on the fp16 shadow of the rows
  bool pre_hi_only = false;   // ... with the query block as fp16 hi only (rows too long for hi + lo in LDS: prescan_kernel_g<.., LO = false>)
  int pre_mode = 1;           // option "prescan"
  uint32_t kp = 0;            // candidate keys per partial list of the matrix-core scan (top_k + slack)
  uint32_t k_keep = 0;        // keys per partial slot
  uint32_t n_pass = 1;        // 64 result ranks per pass (ordered-chain scans)
  uint32_t seg_rows = 0, seg_target = 0, S_max = 0;
  uint64_t items_bound = 0;
  size_t part_bytes = 0;      // partial slots + pruning bounds of the batch
  uint32_t *pj_list = nullptr, *pj_pref = nullptr, *pj_take = nullptr, *np = nullptr, *pj_nq = nullptr;
  uint32_t *cnt = nullptr, *pair_off = nullptr, *group_off = nullptr, *quad_ctr = nullptr, *fail_list = nullptr, *qflags = nullptr;
  GroupTotals* tot = nullptr;
  const float* qp = nullptr;  // staged (padded) queries
  SearchWs::CoarseAhead* took = nullptr;  // the look-ahead slot this search consumed, if any
};
int32_t plan_search(vers_ivf* h, const float* q_dev, uint64_t ldq_in, uint32_t b, uint32_t top_k, uint32_t nprobe, hipStream_t st,
                    SearchPlan& s);

// ---- ivf_search.hip ----------------------------------------------------------------------------------------------------
int32_t search_dev_locked(vers_ivf* h, const float* q_dev, uint64_t ldq_in, uint32_t b, uint32_t top_k, uint32_t nprobe,
                          uint64_t* out_ids, float* out_dist, uint32_t* out_count, uint64_t* out_keys, hipStream_t st);
int32_t upload_queries(const float* queries, uint64_t stride_bytes, uint32_t b, uint32_t d, DevBuf& buf);

// ---- ivf_build.hip -----------------------------------------------------------------------------------------------------
int32_t refresh_norms(vers_ivf* h, uint64_t r_begin, uint64_t r_end, hipStream_t st);
int32_t poison_slack(vers_ivf* h, float value, hipStream_t st);  // test hook / option "poison_slack_bits": rows that hold no vector <- value

}  // namespace ivf
}  // namespace vers
using namespace vers::ivf;
