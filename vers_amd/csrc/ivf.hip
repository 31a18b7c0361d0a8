// ivf.hip -- the IVFFlat index behind vers_ivf_*: build_index / add / search_approximate
// of /root/reference/vers/src/indexes/ivfflat.rs on one MI355X.
//
// Device layout (HBM): the reference keeps `values` in vec_id order and gathers list members
// through `ids` (ivfflat.rs:172-174).  Here rows are stored CLUSTER-MAJOR: list c occupies the
// contiguous rows [list_off[c], list_off[c]+list_len[c]) in the reference's list order
// (ascending vec_id for built rows, append order for added ones), followed by slack for `add`;
// row_ids[] maps a storage row back to its vec_id.  Lists start on 64-row boundaries and the rows
// themselves are held in lane-transposed 64-row tiles (scan.hip.h), so a list scan is one linear HBM
// stream of contiguous 1 KiB wave loads.
//
// search = coarse scan over the centroids (top-P keys) -> plan (which lists, per-query sequence
// bases, reference spill plan) -> group (query,list) pairs by list so a list is streamed once for
// up to 8 queries -> inverted-list scan (scan.hip.h engine) -> per-query merge + id mapping.
// Everything is planned on the device; the host never waits inside a search.
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

#include "gemm.hip.h"
#include "kmeans.hpp"
#include "plan.hip.h"
#include "prescan.hip.h"
#include "scan.hip.h"
#include "util.hip.h"

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

constexpr int32_t kRetrySpill = 1001;  // internal: reference-mode spill ran past the ranked lists, retry deeper if possible

// ---- sources for the scan engine -----------------------------------------------------------
// coarse quantiser / exhaustive scan: item = (row segment, query group), slot(q, seg) = q*n_segs + seg;
// QG > 1: segments padded to a multiple of 4 with empty items (quads share a query block, scan.hip.h)
template <int QG, bool SEQ_IDS>
struct SegSrc {
  static constexpr bool kSeqIds = SEQ_IDS;
  static constexpr bool kStreamOnce = SEQ_IDS;  // exhaustive scan of the stored rows: once; coarse quantiser: centroids are re-read
  const float* rows;
  uint64_t n;
  uint32_t ld;
  uint32_t seg_rows, n_segs, n_segs_pad;
  const float* queries;  // QG == 1: [b][ldq]; else interleaved blocks
  uint32_t ldq, b;
  uint64_t* partials;
  uint32_t k;
  const uint32_t* ids;
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

// inverted-list scan: item = (list, query group of the list, row segment of the list)
template <int QG>
struct IvfSrc {
  static constexpr bool kSeqIds = false;
  static constexpr bool kStreamOnce = true;
  const float* rows;
  uint32_t ld;
  const uint32_t* list_off;  // storage row of each (local) list
  const uint32_t* list_len;
  const ItemDesc* items;
  const uint32_t* n_items_dev;
  const uint32_t* cnt;        // pairs per list
  const uint32_t* pair_off;   // first pair of each list
  const uint32_t* pairs;      // pair -> q*P + j
  const uint32_t* group_off;  // first group id of each list
  const float* qblocks;       // QG > 1: [group][ldq][QG]
  const float* qp;            // QG == 1: padded queries [b][ldq]
  uint32_t ldq, P, S_max, k_keep, seg_rows, seg_target;
  uint32_t bound_per_pair;    // reference mode merges per (query, list); nprobe mode per query
  const uint32_t* pj_pref;    // [b*P] sequence base of probe j of query q
  uint64_t* partials;         // [b*P*S_max][k_keep]

  __device__ __forceinline__ uint32_t n_items() const { return *n_items_dev; }
  __device__ __forceinline__ void get(uint32_t it, ItemView<QG>& v) const {
    const ItemDesc d = items[it];
    const bool real = d.seg != kNoSeg;
    const uint32_t len = list_len[d.list];
    const uint32_t sr = list_seg_rows(len, seg_rows, seg_target);
    const uint32_t r0 = real ? d.seg * sr : 0;
    v.rows = rows + ((uint64_t)list_off[d.list] + r0) * ld;
    v.row0 = r0;
    v.nrows = real ? (len - r0 < sr ? len - r0 : sr) : 0u;
    const uint32_t c = cnt[d.list] - d.group * QG;
    v.nq = c < (uint32_t)QG ? c : QG;
    if (QG == 1) v.qb = qp + (uint64_t)(pairs[pair_off[d.list] + d.group] / P) * ldq;
    else v.qb = qblocks + (uint64_t)(group_off[d.list] + d.group) * ldq * QG;
  }
  __device__ __forceinline__ uint32_t pair_of(uint32_t it, int qi) const {
    const ItemDesc d = items[it];
    return pairs[pair_off[d.list] + d.group * QG + qi];
  }
  __device__ __forceinline__ const float* query_row(uint32_t it, uint32_t qi) const {  // padded query of slot qi
    const ItemDesc d = items[it];
    return qp + (uint64_t)(pairs[pair_off[d.list] + d.group * QG + qi] / P) * ldq;
  }
  __device__ __forceinline__ uint32_t storage_row(uint32_t it) const {  // first storage row of the item
    const ItemDesc d = items[it];
    return list_off[d.list] + (d.seg != kNoSeg ? d.seg * list_seg_rows(list_len[d.list], seg_rows, seg_target) : 0u);
  }
  __device__ __forceinline__ uint32_t seq_base(uint32_t it, int qi) const {
    const ItemDesc d = items[it];
    return pj_pref[pairs[pair_off[d.list] + d.group * QG + qi]] + d.seg * list_seg_rows(list_len[d.list], seg_rows, seg_target);
  }
  __device__ __forceinline__ const uint32_t* seq_ids(uint32_t) const { return nullptr; }
  __device__ __forceinline__ uint64_t* out(uint32_t it, int qi) const {  // ordered-chain scan: S_max counts segments
    return partials + ((uint64_t)pair_of(it, qi) * S_max + items[it].seg) * k_keep;
  }
  // slot of a whole quad of segments (matrix-core scan: one list per query and block); S_max counts quads there
  __device__ __forceinline__ uint64_t* out_quad(uint32_t it0, int qi) const {
    return partials + ((uint64_t)pair_of(it0, qi) * S_max + (items[it0].seg >> 2)) * k_keep;
  }
  __device__ __forceinline__ uint32_t bound_slot(uint32_t it, int qi) const {
    const uint32_t pr = pair_of(it, qi);
    return bound_per_pair ? pr : pr / P * P;
  }
  __device__ __forceinline__ uint32_t slot_of_pair(uint32_t pr) const { return bound_per_pair ? pr : pr / P * P; }
};

// Single query (planned by plan1_block): an item is a 16-byte RECORD -- where its rows are, how many, the sequence number of the
// first and its partial slot -- so that a wave reaches its first tile load after ONE round trip.  Through IvfSrc it is five
// dependent ones (item count -> item -> the list's tables -> pair -> sequence base): ~3 us of a 58 us launch in which every wave
// has exactly one item.
struct Item1Rec {
  uint32_t row0;   // storage row of the item's first row
  uint32_t nrows;
  uint32_t seq0;   // sequence number of the first row (the probe's base + the segment's offset in its list)
  uint32_t out;    // partial slot, in units of k_keep keys: pair * S_max + segment
};
struct Rec1Src {  // what scan_item asks of its source, answered from the record in registers
  static constexpr bool kSeqIds = false;
  static constexpr bool kStreamOnce = true;
  Item1Rec r;
  uint64_t* partials;
  uint32_t k_keep, S_max, bound_per_pair;
  __device__ __forceinline__ uint32_t seq_base(uint32_t, int) const { return r.seq0; }
  __device__ __forceinline__ const uint32_t* seq_ids(uint32_t) const { return nullptr; }
  __device__ __forceinline__ uint64_t* out(uint32_t, int) const { return partials + (uint64_t)r.out * k_keep; }
  __device__ __forceinline__ uint32_t bound_slot(uint32_t, int) const { return bound_per_pair ? r.out / S_max : 0u; }
};
struct Scan1Args {
  const float* rows; const Item1Rec* recs; const uint32_t* n_items_dev; const float* qp;
  uint64_t* partials; uint32_t k_keep, S_max, bound_per_pair;
};
template <int METRIC>
__global__ __launch_bounds__(kWave * kWavesPerBlock) void scan1_kernel(Scan1Args a, ScanParams p) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t n_waves = gridDim.x * kWavesPerBlock;
  uint32_t it = blockIdx.x * kWavesPerBlock + wid;
  // (both loads go out together: the record buffer holds at least one entry per launched wave, a stale one is never used)
  const uint32_t n_items = *a.n_items_dev;
  Rec1Src src;
  src.r = a.recs[it];
  src.partials = a.partials; src.k_keep = a.k_keep; src.S_max = a.S_max; src.bound_per_pair = a.bound_per_pair;
  bool nan_seen = false;
  while (it < n_items) {
    ItemView<1> v;
    v.rows = a.rows + (uint64_t)src.r.row0 * p.ld;
    v.nrows = src.r.nrows;
    v.nq = 1;
    v.qb = a.qp;
    scan_item<1, 1, METRIC>(src, p, it, v, lane, nan_seen);
    it += n_waves;
    if (it < n_items) src.r = a.recs[it];
  }
  if (__ballot(nan_seen) != 0 && lane == 0) atomicOr(p.status, 1u);
}

// ---- small kernels of the search pipeline -----------------------------------------------------
// coarse merge: one block per query, top-P centroid keys (ascending (dist, centroid index)).  P > 64: 64 ranks per pass
// (ScanParams::lower) -- this pass's keys go to probe[q][rank0 ..], its last key becomes the next pass's lower bound.
__global__ __launch_bounds__(kWave * kMergeWaves) void coarse_merge_kernel(const uint64_t* partials, uint32_t n_segs, uint32_t k_pass,
                                                                           uint32_t P, uint32_t rank0, uint64_t* probe, uint64_t* lower_out) {
  __shared__ uint64_t sh[kMergeWaves][kWave];
  const uint32_t q = blockIdx.x;
  uint64_t list = block_merge_keys(partials + (uint64_t)q * n_segs * k_pass, n_segs * k_pass, k_pass, sh);
  if (threadIdx.x < k_pass) probe[(uint64_t)q * P + rank0 + threadIdx.x] = list;
  if (lower_out != nullptr && threadIdx.x == k_pass - 1) lower_out[q] = list;  // (kKeyMax when the centroids ran out: the next pass finds nothing)
}

// ---- planning of a batch (plan.hip.h): no cross-block waiting anywhere -------------------------------------------
// Step (1) standalone: a wave per query reads its ranked lists from `probe` (exact coarse quantiser, a look-ahead slot,
// more than 64 ranked lists).  Batches ranked on the matrix cores get this step in the tail of the selection kernel.
__global__ __launch_bounds__(256) void plan_queries_kernel(PlanQ a, const uint64_t* probe) {
  const uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= a.b) return;  // (whole waves)
  uint32_t carry = 0, n_visited = 0;
  for (uint32_t c0 = 0; c0 < a.P; c0 += kWave) {  // lane j of chunk c = probe rank 64c + j; the running row count carries over
    const uint32_t j = c0 + (uint32_t)lane;
    plan_query_chunk(a, q, lane, c0, j < a.P ? probe[(uint64_t)q * a.P + j] : kKeyMax, carry, n_visited);
  }
  plan_query_finish(a, q, lane, carry, n_visited);
}

// Steps (2) + (3): ONE ordinary launch of up to 64 blocks.  EVERY block runs the prefix sums over all lists (per list:
// pairs, groups = ceil(cnt / QG), items = groups * segments; a few microseconds of L2 reads) and keeps / stores the
// entries of the lists it OWNS -- granules of four consecutive slots dealt round-robin over the blocks -- then scatters
// the pairs of its lists and writes their item / group descriptors.  Nothing waits for another block.
// All tables are in SLOT order (lists by descending length, see vers_ivf::list_slot).  Work order of the scan = hot
// lists first (nearest list of some query: their thresholds must be tight before the bulk is scanned), then the others
// in slot order, i.e. LONGEST FIRST.
constexpr uint32_t kGroupThreads = 1024, kGroupMaxBlocks = 64;
struct GroupArgs {
  uint32_t b, P, k_lists, QG, seg_rows, seg_target;
  const uint32_t* slot_len;   // list lengths in slot order
  const uint32_t* cnt;        // pairs per list (plan_query)
  const uint32_t* hot;
  uint32_t* fill;             // zeroed with cnt
  const uint32_t* pj_list;
  uint32_t *pair_off, *group_off, *item_off;
  GroupTotals* tot;
  uint32_t* pairs;
  ItemDesc* items;
  GroupDesc* groups;
  u32x4* ff_begin;       // 0xFF-filled here: the pruning bounds (matrix-core scan; its slots need no fill: ivf_rescore_kernel
  uint64_t ff_vec16;     // reads written slots only).  This many 16-byte words.
  unsigned long long* stamps;  // diagnosis (VERS_SCAN_DEBUG & 16): [16..19] 100 MHz clock at the phase boundaries, block 0
};
// a table entry this block stored itself a phase ago: read past the vector L1 (which may hold the line from before the store)
__device__ __forceinline__ uint32_t ld_l2(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool owns_list(uint32_t L) { return ((L >> 2) % gridDim.x) == blockIdx.x; }

// Four exclusive prefix sums over the lists (pairs, groups, items of hot lists, items of the others) in rounds of 4096
// lists: a thread owns FOUR consecutive lists (three 16-byte loads, a serial scan in registers), the waves scan the
// thread totals by shuffles, 16 wave totals go through LDS, a running carry links the rounds.  One round and two block
// barriers at 4096 lists.
__device__ __forceinline__ uint32_t group_lists(const GroupArgs& a, uint32_t* tab) {
  __shared__ uint32_t wp[16], wg[16], wi[16], wh[16];
  __shared__ unsigned long long ur, sr;
  const uint32_t* cnt = a.cnt; const uint32_t* list_len = a.slot_len; const uint32_t* hot = a.hot;
  const uint32_t k_lists = a.k_lists, QG = a.QG;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) { ur = sr = 0; }
  uint32_t cp = 0, cg = 0, ci = 0, ch = 0;  // carries (identical in every thread); ci: other lists' items, ch: hot lists' items
  unsigned long long my_ur = 0, my_sr = 0;
  auto wave_incl = [&](uint32_t x) {
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const uint32_t t = __shfl_up(x, off, kWave);
      if (lane >= off) x += t;
    }
    return x;
  };
  // whole 16-byte vectors when every table starts on one (they are carved out of one allocation: true when k, b are multiples of 4)
  const bool vec_ok = (k_lists & 3u) == 0 &&
                      (((uintptr_t)cnt | (uintptr_t)list_len | (uintptr_t)hot | (uintptr_t)a.pair_off | (uintptr_t)a.group_off | (uintptr_t)a.item_off) & 15u) == 0;
  for (uint32_t base0 = 0; base0 < k_lists; base0 += 4 * kGroupThreads) {
    const uint32_t i0 = base0 + 4u * threadIdx.x;
    uint32_t c4[4] = {0, 0, 0, 0}, l4[4] = {0, 0, 0, 0}, h4[4] = {0, 0, 0, 0};
    if (vec_ok && i0 < k_lists) {
      const u32x4 cv = *reinterpret_cast<const u32x4*>(cnt + i0), lv = *reinterpret_cast<const u32x4*>(list_len + i0),
                  hv = *reinterpret_cast<const u32x4*>(hot + i0);
#pragma unroll
      for (int e = 0; e < 4; ++e) { c4[e] = cv[e]; l4[e] = lv[e]; h4[e] = hv[e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (i0 + e < k_lists) { c4[e] = cnt[i0 + e]; l4[e] = list_len[i0 + e]; h4[e] = hot[i0 + e]; }
    }
    uint32_t g4[4], ic4[4], ih4[4];
    uint32_t tp = 0, tg = 0, tic = 0, tih = 0;  // the thread's totals
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      uint32_t g = 0, it = 0;
      if (c4[e]) {
        g = (c4[e] + QG - 1) / QG;
        const uint32_t sr2 = list_seg_rows(l4[e], a.seg_rows, a.seg_target);
        const uint32_t n_s = (l4[e] + sr2 - 1) / sr2;
        it = g * (QG == 1 ? n_s : (n_s + 3) / 4 * 4);  // QG > 1: quads of items share a query block
        my_ur += l4[e];
        my_sr += (unsigned long long)l4[e] * g;
      }
      g4[e] = g; ic4[e] = h4[e] ? 0u : it; ih4[e] = h4[e] ? it : 0u;
      tp += c4[e]; tg += g; tic += ic4[e]; tih += ih4[e];
    }
    const uint32_t ip = wave_incl(tp), ig = wave_incl(tg), ii = wave_incl(tic), ih = wave_incl(tih);
    __syncthreads();  // previous round's readers of wp/wg/wi/wh are done
    if (lane == kWave - 1) { wp[wid] = ip; wg[wid] = ig; wi[wid] = ii; wh[wid] = ih; }
    __syncthreads();
    uint32_t bp = 0, bg = 0, bi2 = 0, bh = 0, rp = 0, rg = 0, ri = 0, rh = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < wid) { bp += wp[w]; bg += wg[w]; bi2 += wi[w]; bh += wh[w]; }
      rp += wp[w]; rg += wg[w]; ri += wi[w]; rh += wh[w];
    }
    // exclusive offsets of the thread's first list, then along its four
    uint32_t op = cp + bp + ip - tp, og = cg + bg + ig - tg, oc = ci + bi2 + ii - tic, oh = ch + bh + ih - tih;
    uint32_t po[4], go[4], io[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      po[e] = op; go[e] = og; io[e] = h4[e] ? oh : oc;  // (the others' item offsets are shifted behind the hot lists' below)
      op += c4[e]; og += g4[e]; oc += ic4[e]; oh += ih4[e];
    }
    if (tab != nullptr) {  // first pair of EVERY list, block-local: the scatter below is then dealt by pair, not by list;
#pragma unroll         // first group / item of the lists this block describes
      for (int e = 0; e < 4; ++e)
        if (i0 + e < k_lists) { tab[i0 + e] = po[e]; tab[k_lists + i0 + e] = go[e]; tab[2 * k_lists + i0 + e] = io[e]; }
    }
    if (i0 < k_lists && owns_list(i0)) {  // the granule's owner stores its entries
      if (vec_ok) {
        *reinterpret_cast<u32x4*>(a.pair_off + i0) = u32x4{po[0], po[1], po[2], po[3]};
        *reinterpret_cast<u32x4*>(a.group_off + i0) = u32x4{go[0], go[1], go[2], go[3]};
        *reinterpret_cast<u32x4*>(a.item_off + i0) = u32x4{io[0], io[1], io[2], io[3]};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (i0 + e < k_lists) { a.pair_off[i0 + e] = po[e]; a.group_off[i0 + e] = go[e]; a.item_off[i0 + e] = io[e]; }
      }
    }
    cp += rp; cg += rg; ci += ri; ch += rh;
  }
  // (the item offsets of the lists that are not hot are stored UNSHIFTED: their only reader, list_items below, adds the hot
  // lists' total `ch` -- identical in every thread -- itself; round 2 read every entry back and rewrote it)
  if (blockIdx.x == 0) {  // traffic statistics of the batch: one block's job
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {  // (1024 threads adding into two LDS words one by one was most of this kernel)
      my_ur += __shfl_xor(my_ur, off, kWave);
      my_sr += __shfl_xor(my_sr, off, kWave);
    }
    if (lane == 0) {
      atomicAdd(&ur, my_ur);
      atomicAdd(&sr, my_sr);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      a.tot->n_items = ci + ch; a.tot->n_groups = cg; a.tot->n_pairs = cp; a.tot->pad = 0;
      a.tot->union_rows = ur; a.tot->streamed_rows = sr;
    }
  }
  return ch;
}

// items + group descriptors of one list (of this block: the offsets are its own stores of the phase before)
__device__ __forceinline__ void list_items(uint32_t L, const GroupArgs& a, const uint32_t* tab, uint32_t hot_items) {
  const uint32_t c = a.cnt[L];
  if (!c) return;
  const uint32_t QG = a.QG, len = a.slot_len[L];
  const uint32_t sr = list_seg_rows(len, a.seg_rows, a.seg_target);
  const uint32_t n_g = (c + QG - 1) / QG, n_s = (len + sr - 1) / sr;
  const uint32_t n_s_pad = QG == 1 ? n_s : (n_s + 3) / 4 * 4;
  uint32_t o = (tab ? tab[2 * a.k_lists + L] : ld_l2(a.item_off + L)) + (a.hot[L] ? 0u : hot_items);  // work order: hot lists first
  const uint32_t g0 = tab ? tab[a.k_lists + L] : ld_l2(a.group_off + L), p0 = tab ? tab[L] : ld_l2(a.pair_off + L);
  for (uint32_t g = 0; g < n_g; ++g) a.groups[g0 + g] = GroupDesc{p0 + g * QG, (c - g * QG < QG) ? c - g * QG : QG};
  if (QG == 1) {
    for (uint32_t g = 0; g < n_g; ++g)
      for (uint32_t s = 0; s < n_s; ++s) a.items[o++] = ItemDesc{L, g, s};
  } else {
    // quads of segments outermost, query groups inside: the groups that re-read the same rows are
    // neighbours in the item order, and scan_kernel's XCD remap runs neighbours on one XCD's L2
    for (uint32_t s0 = 0; s0 < n_s_pad; s0 += 4)
      for (uint32_t g = 0; g < n_g; ++g)
        for (uint32_t s = s0; s < s0 + 4; ++s) a.items[o++] = ItemDesc{L, g, s < n_s ? s : kNoSeg};
  }
}

constexpr uint32_t kGroupTabMax = 8192;  // lists whose three offset tables a block keeps in LDS (96 KB)
__global__ __launch_bounds__(kGroupThreads) void group_scatter_kernel(GroupArgs a) {
  extern __shared__ uint32_t pair_tab[];  // first pair | group | item of every list: [3][k_lists] when k_lists <= kGroupTabMax
  const uint32_t tid = blockIdx.x * kGroupThreads + threadIdx.x, nthreads = gridDim.x * kGroupThreads;
  auto stamp = [&](int i) { if (a.stamps && tid == 0) a.stamps[16 + i] = __builtin_amdgcn_s_memrealtime(); };
  stamp(0);
  const bool use_tab = a.k_lists <= kGroupTabMax;
  const u32x4 ff = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  for (uint64_t i = tid; i < a.ff_vec16; i += nthreads) a.ff_begin[i] = ff;
  const uint32_t hot_items = group_lists(a, use_tab ? pair_tab : nullptr);
  // what a block reads back below it stored ITSELF: its stores only have to have reached ITS L2 (release at workgroup
  // scope = wait for them; an agent-scope fence writes the whole L2 back -- that alone was 8 us here) and the read-backs
  // go past the vector L1 (ld_l2).  With the LDS tables nothing is read back at all.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  stamp(1);
  // pairs of a list become contiguous (order inside a list is arbitrary and irrelevant: every (query, list) result goes
  // to its own slot)
  const uint32_t n_pj = a.b * a.P;
  if (use_tab) {  // every block knows every list's first pair: the pairs are dealt over ALL threads of the grid
    for (uint32_t i = tid; i < n_pj; i += nthreads) {
      const uint32_t L = a.pj_list[i];
      if (L != kNoList) a.pairs[pair_tab[L] + atomicAdd(&a.fill[L], 1u)] = i;
    }
  } else {  // more lists than the table holds: each block picks the pairs of ITS lists out of the whole table (8 independent loads a time)
    for (uint32_t i0 = threadIdx.x; i0 < n_pj; i0 += 8 * kGroupThreads) {
      uint32_t Ls[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) Ls[u] = i0 + u * kGroupThreads < n_pj ? a.pj_list[i0 + u * kGroupThreads] : kNoList;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (Ls[u] != kNoList && owns_list(Ls[u])) a.pairs[ld_l2(a.pair_off + Ls[u]) + atomicAdd(&a.fill[Ls[u]], 1u)] = i0 + u * kGroupThreads;
    }
  }
  stamp(2);
  // item and group descriptors of its lists: granule g = blockIdx.x + n * gridDim.x, four lists each
  const uint32_t n_gran = (a.k_lists + 3) / 4;
  for (uint32_t w = threadIdx.x; ; w += kGroupThreads) {
    const uint32_t gran = blockIdx.x + (w >> 2) * gridDim.x;
    if (gran >= n_gran) break;
    const uint32_t L = 4 * gran + (w & 3u);
    if (L < a.k_lists) list_items(L, a, use_tab ? pair_tab : nullptr, hot_items);
  }
  stamp(3);
}

// Single query: coarse merge + plan + group + scatter + items by ONE block (the five small kernels above cost
// ~20 us each in launch + latency, more than the 60 us list scan they prepare): 16 waves merge the coarse partial
// slots, wave 0 then plans with lane j = probe rank j.  Every probed list is distinct here, so a pair is its own
// group; only the table entries of probed lists are written (no memset of the per-list arrays).
struct Plan1Args {
  const uint64_t* cpart; uint32_t n_segs_c, P, k_lists, top_k; int ref_mode;
  const uint32_t* list_len; const uint8_t* owner; uint32_t rank, seg_rows;
  uint64_t* probe; uint32_t *pj_list, *pj_pref, *pj_take, *np, *cnt, *pair_off, *group_off, *pairs;
  ItemDesc* items; GroupDesc* groups; GroupTotals* tot; uint32_t* status; const uint32_t* list_slot;
  Item1Rec* recs = nullptr; const uint32_t *slot_off = nullptr, *slot_len = nullptr; uint32_t S_max = 0;  // recs != nullptr: the items as records (scan1_kernel)
  u32x4* ff_begin; uint32_t ff_vec16;  // the list scan's partial slots: filled with all ones (empty) by whoever plans
  unsigned long long* stamps = nullptr;  // diagnosis (VERS_SCAN_DEBUG & 16): [48..50] 100 MHz clock after the merge, the list tables, the plan's stores
};
template <bool COHERENT>  // the coarse slots come from other blocks of this launch (coarse1_kernel)
__device__ __forceinline__ void plan1_block(const Plan1Args& a, uint64_t (*sh)[kWave]) {
  const uint64_t list = block_merge_keys<kMergeWaves, MergeNoOp, COHERENT>(a.cpart, a.n_segs_c * a.P, a.P, sh);
  if (threadIdx.x >= kWave) return;
  const int lane = threadIdx.x;
  if (a.stamps && lane == 0) a.stamps[48] = __builtin_amdgcn_s_memrealtime();
  const uint32_t P = a.P, top_k = a.top_k;
  const uint64_t key = lane < (int)P ? list : kKeyMax;
  if (lane < (int)P) a.probe[lane] = key;
  const uint32_t L = key != kKeyMax ? (uint32_t)key : kNoList;
  const uint32_t len = L != kNoList ? a.list_len[L] : 0u;
  const uint32_t slot = L != kNoList ? a.list_slot[L] : kNoList;  // the tables and items name a list by its slot (vers_ivf::list_slot)
  // (the records' operands depend on the slot: requested here, they arrive under the prefix sums below)
  const uint32_t loff = (a.recs && slot != kNoList) ? a.slot_off[slot] : 0u, llen = (a.recs && slot != kNoList) ? a.slot_len[slot] : 0u;
  auto excl_scan = [&](uint32_t v) {  // exclusive prefix sum over the 64 lanes
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const uint32_t t = __shfl_up(inc, off, kWave);
      if (lane >= off) inc += t;
    }
    return inc - v;
  };
  const uint32_t pref = excl_scan(len);
  if (a.stamps && lane == 0) a.stamps[49] = __builtin_amdgcn_s_memrealtime();
  const uint32_t total_rows = (uint32_t)__shfl(pref + len, kWave - 1, kWave);
  // reference mode (ivfflat.rs:166-195) in closed form: list j is visited while the rows before it do not yet
  // fill top_k, and contributes take_j = min(len_j, top_k - rows before it)
  const bool visited = L != kNoList && (!a.ref_mode || pref < top_k);
  const uint32_t take = !visited ? 0u : (a.ref_mode ? (len < top_k - pref ? len : top_k - pref) : top_k);
  const bool scan = visited && len > 0 && take > 0 && (a.owner == nullptr || a.owner[L] == a.rank);
  if (lane < (int)P) {
    a.pj_list[lane] = scan ? slot : kNoList;
    a.pj_pref[lane] = pref;
    a.pj_take[lane] = take;
  }
  const uint64_t vmask = __ballot(visited), smask = __ballot(scan);
  if (lane == 0) {
    a.np[0] = (uint32_t)__popcll(vmask);
    if (a.ref_mode && top_k > 0 && total_rows < top_k) atomicOr(a.status, P >= a.k_lists ? kStInsufficient : kStSpillTooDeep);
  }
  const uint32_t n_s = scan ? (len + a.seg_rows - 1) / a.seg_rows : 0u;
  const uint32_t item0 = excl_scan(n_s);
  const uint32_t pidx = (uint32_t)__popcll(smask & ((1ull << lane) - 1ull));
  if (scan) {
    a.cnt[slot] = 1; a.pair_off[slot] = pidx; a.group_off[slot] = pidx;
    a.pairs[pidx] = (uint32_t)lane;  // q*P + j with q = 0
    a.groups[pidx] = GroupDesc{pidx, 1u};
    if (a.recs) {
      // (llen: the stored length, what IvfSrc::get cuts the segments from)
      for (uint32_t sgi = 0; sgi < n_s; ++sgi) {
        const uint32_t r0 = sgi * a.seg_rows;
        a.recs[item0 + sgi] = Item1Rec{loff + r0, llen - r0 < a.seg_rows ? llen - r0 : a.seg_rows, pref + r0, (uint32_t)lane * a.S_max + sgi};
      }
    } else {
      for (uint32_t sgi = 0; sgi < n_s; ++sgi) a.items[item0 + sgi] = ItemDesc{slot, 0u, sgi};
    }
  }
  if (a.stamps && lane == 0) a.stamps[50] = __builtin_amdgcn_s_memrealtime();
  const uint32_t n_items = (uint32_t)__shfl(item0 + n_s, kWave - 1, kWave);
  const uint32_t rows_scanned = (uint32_t)__shfl(excl_scan(scan ? len : 0u) + (scan ? len : 0u), kWave - 1, kWave);
  if (lane == 0) {
    a.tot->n_items = n_items; a.tot->n_groups = (uint32_t)__popcll(smask); a.tot->n_pairs = a.tot->n_groups; a.tot->pad = 0;
    a.tot->union_rows = rows_scanned; a.tot->streamed_rows = rows_scanned;
  }
}
__global__ __launch_bounds__(kWave * kMergeWaves) void plan1_kernel(Plan1Args a) {
  __shared__ uint64_t sh[kMergeWaves][kWave];
  {  // the block's 1024 threads fill the slots here instead of a memset launch of its own
    const u32x4 ff = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (uint32_t i = threadIdx.x; i < a.ff_vec16; i += kWave * kMergeWaves) a.ff_begin[i] = ff;
  }
  plan1_block<false>(a, sh);
}

// Single query, coarse quantiser (ivfflat.rs:155-161) AND the plan in one launch.  The ordered-chain scan gives a 64-centroid
// tile to ONE wave, which walks its 192 KiB with 24 KiB in flight: eight dependent round trips, 24 us for 12.6 MB that sit in
// the caches.  Here a tile belongs to a BLOCK of 16 waves: wave w loads chunk w (32 columns of the 64 rows) of every phase of
// 16 chunks -- the whole tile is in flight at once -- and computes its rows' PRODUCTS (x - q)^2 (or x * q), which do not depend
// on the running sum, into LDS; wave 0 then walks the strictly ordered chain acc = acc + m_j over the products: the same
// operations on the same operands in the same order as scan_item's chain (base.rs:119-126), one dependent add per column
// instead of three instructions.  The block that finishes last (a device counter; nobody waits for anybody) merges the
// tiles' slots and plans (plan1_block).
struct Coarse1Args {
  const float* cent;  // lane-transposed 64-row tiles
  uint32_t k, ld, n_chunks;
  const float* q;     // the query, zero padded to ld
  uint64_t* cpart;    // [tiles][P]
  uint32_t P;
  uint32_t* ctr;      // zero between launches (the last block resets it)
  uint32_t* status;
  unsigned long long* stamps;  // diagnosis (VERS_SCAN_DEBUG & 16): [32..39] 100 MHz clock at the phase boundaries of the LAST block
};
constexpr int kC1Phase = 16;  // chunks of a phase = waves of the block
constexpr size_t kC1LdsBytes = (size_t)kC1Phase * kLoads * kWave * sizeof(f32x4);  // 128 KiB of products
static_assert(kC1Phase == kMergeWaves, "the planning tail needs the merge's 16 waves");
static_assert(kC1LdsBytes >= sizeof(uint64_t) * kMergeWaves * kWave, "the merge's exchange area reuses the product buffer");
template <int METRIC>
__global__ __launch_bounds__(kWave * kC1Phase) void coarse1_kernel(Coarse1Args c, Plan1Args a) {
  extern __shared__ __attribute__((aligned(16))) f32x4 prod[];  // [chunk of the phase][load][lane]
  __shared__ uint32_t s_last;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t tile = blockIdx.x;
  unsigned long long ts[5] = {};
  auto stamp = [&](int i) { if (c.stamps) ts[i] = __builtin_amdgcn_s_memrealtime(); };
  stamp(0);
  {  // the list scan's partial slots start out empty: every block fills its share while its loads are in flight
    const u32x4 ff = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (uint32_t i = blockIdx.x * (kWave * kC1Phase) + threadIdx.x; i < a.ff_vec16; i += gridDim.x * (kWave * kC1Phase)) a.ff_begin[i] = ff;
  }
  const uint32_t tile_bytes = c.ld * 256u;
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)(c.cent + (uint64_t)tile * kWave * c.ld), 0, (int)tile_bytes, 0x00020000);
  const uint32_t lane_off = (uint32_t)lane * 16u;
  // (chunks past the end of the tile are out of the descriptor's range: they load zeros and are never used)
  auto issue = [&](u32x4 (&r)[kLoads], uint32_t ch) {
#pragma unroll
    for (int i = 0; i < kLoads; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, ch * (kLoads * 1024u) + (uint32_t)i * 1024u, 0);
  };
  auto products = [&](const u32x4 (&r)[kLoads], uint32_t ch) {
    cfloat_as4* qs = (cfloat_as4*)(c.q + ch * kChunk);
#pragma unroll
    for (int i = 0; i < kLoads; ++i) {
      f32x4 m;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float xv = __uint_as_float(r[i][u]);
        const float sv = qs[i * 4 + u];
        if (METRIC == 0) {
          const float t = __fsub_rn(xv, sv);
          m[u] = __fmul_rn(t, t);
        } else {
          m[u] = __fmul_rn(xv, sv);
        }
      }
      prod[(wid * kLoads + i) * kWave + lane] = m;
    }
  };
  u32x4 bufA[kLoads], bufB[kLoads];
  float acc = 0.0f;
  auto phase = [&](const u32x4 (&cur)[kLoads], u32x4 (&nxt)[kLoads], uint32_t c0) {
    issue(nxt, c0 + kC1Phase + (uint32_t)wid);  // the next phase's chunk: in flight under this phase's chain
    if (c0 + (uint32_t)wid < c.n_chunks) products(cur, c0 + (uint32_t)wid);
    __syncthreads();
    if (c0 == 0) stamp(1);
    if (wid == 0) {
      const uint32_t nch = c.n_chunks - c0 < (uint32_t)kC1Phase ? c.n_chunks - c0 : (uint32_t)kC1Phase;
      auto ld = [&](f32x4 (&m)[kLoads], uint32_t s) {
#pragma unroll
        for (int i = 0; i < kLoads; ++i) m[i] = prod[(s * kLoads + i) * kWave + lane];
      };
      auto add = [&](const f32x4 (&m)[kLoads]) {
#pragma unroll
        for (int i = 0; i < kLoads; ++i)
#pragma unroll
          for (int u = 0; u < 4; ++u) acc = __fadd_rn(acc, m[i][u]);
      };
      // (two register buffers: the next chunk's products are on their way from LDS while this chunk's 32 adds run.  The chain itself is
      // the floor: a dependent v_add_f32 issues every ~8 cycles, 768 of them are ~3 us; fully unrolled phases with the reads pinned two
      // half-chunks ahead measured 3.5 us against this loop's 3.7)
      f32x4 mA[kLoads], mB[kLoads];
      const uint32_t last = nch - 1;
      ld(mA, 0);
      for (uint32_t s = 0; s < nch; s += 2) {
        ld(mB, s + 1 < nch ? s + 1 : last);
        add(mA);
        if (s + 1 < nch) {
          ld(mA, s + 2 < nch ? s + 2 : last);
          add(mB);
        }
      }
    }
    __syncthreads();
  };
  issue(bufA, (uint32_t)wid);
  for (uint32_t c0 = 0; c0 < c.n_chunks; c0 += 2 * kC1Phase) {
    phase(bufA, bufB, c0);
    if (c0 + kC1Phase < c.n_chunks) phase(bufB, bufA, c0 + kC1Phase);
  }
  stamp(2);
  if (wid == 0) {
    const uint32_t row = tile * kWave + (uint32_t)lane;
    const bool valid = row < c.k;
    const float dist = METRIC == 0 ? acc : __fsub_rn(1.0f, acc);
    if (__ballot(valid && dist != dist) != 0 && lane == 0) atomicOr(c.status, kStNaN);
    uint64_t key = valid ? make_key(dist, row) : kKeyMax;
    wave_bitonic_sort64(key, lane);
    // The slot is read by ANOTHER BLOCK OF THE SAME LAUNCH: it is stored at agent scope -- written through this XCD's L2 -- so
    // that no L2 write-back (the release fence at agent scope: measured 2 us here with 8 blocks per XCD, 47 us over the 368
    // blocks of a single-query list scan) is needed; once the stores have COMPLETED the block counts itself finished.
    // A workgroup-scope release does not wait for global stores on gfx950 (only lgkmcnt): the wait is spelled out -- vmcnt(0)
    // retires the write-through stores (they are acknowledged by the L2 they were written through to) before the counter moves.
    // (Round 3 shipped without it: store, s_waitcnt lgkmcnt(0), atomic add -- the last block could have merged a stale slot.)
    if (lane < (int)c.P) __hip_atomic_store(c.cpart + (uint64_t)tile * c.P + lane, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) s_last = __hip_atomic_fetch_add(c.ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == gridDim.x ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  stamp(3);
  // (no acquire fence: it would invalidate this XCD's L2 and send the plan's table loads to memory -- 2 x 1.5 us of dependent
  // round trips; the other blocks' slots are read at agent scope instead, block_merge_keys<.., COHERENT>)
  if (threadIdx.x == 0) *c.ctr = 0u;  // (the next launch on this workspace is ordered behind this one)
  stamp(4);
  plan1_block<true>(a, reinterpret_cast<uint64_t(*)[kWave]>(prod));
  if (c.stamps && threadIdx.x == 0) {
    for (int i = 0; i < 5; ++i) c.stamps[32 + i] = ts[i];
    c.stamps[37] = __builtin_amdgcn_s_memrealtime();
    c.stamps[38] = blockIdx.x;
  }
}

// interleaved query block of every group: qblocks[(g*ldq + col)*QG + qi]
// (ordered-chain batched scans only: the matrix-core scan gathers its query block from qp while staging it)
__global__ void gather_qblocks_kernel(const GroupDesc* groups, const GroupTotals* tot, const uint32_t* pairs, uint32_t P,
                                      const float* qp, uint32_t ldq, uint32_t QG, float* qblocks) {
  const uint32_t g = blockIdx.x;
  if (g >= tot->n_groups) return;
  const GroupDesc gd = groups[g];
  for (uint32_t i = threadIdx.x; i < ldq * QG; i += blockDim.x) {
    const uint32_t qi = i % QG, col = i / QG;
    float v = 0.0f;
    if (qi < gd.nq) v = qp[(uint64_t)(pairs[gd.pair_start + qi] / P) * ldq + col];
    qblocks[(uint64_t)g * ldq * QG + i] = v;
  }
}

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
  if (m.st_host && threadIdx.x == 0) *m.st_host = __hip_atomic_load(m.st_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
__global__ __launch_bounds__(kWave) void rank_merge_kernel(const uint64_t* keys, const uint64_t* ids, uint64_t rank_stride,
                                                           uint32_t world, uint32_t b, uint32_t k, int ref_mode,
                                                           uint64_t* out_ids, float* out_dist, uint32_t* out_count) {
  const uint32_t q = blockIdx.x;
  const int lane = threadIdx.x;
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

// ---- storage construction ---------------------------------------------------------------------
// rows of X (vec_id order, row-major pitch ldx) -> cluster-major storage in lane-transposed tiles;
// grid-stride over (sorted position, float4 column)
// (columns >= d of X are the caller's padding and may hold anything: they are stored as zeros)
__global__ void gather_rows_kernel(const float* X, uint32_t ldx, uint32_t d, uint32_t ld, const uint32_t* sorted_ids,
                                   const uint32_t* assign, const uint32_t* starts, const uint32_t* list_off,
                                   const uint8_t* owner, uint32_t rank, uint64_t n, float* rows, uint32_t* row_ids) {
  const uint32_t ld4 = ld / 4, ldx4 = ldx / 4;
  const uint64_t total = n * ld4;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t p = i / ld4;
    const uint32_t c4 = (uint32_t)(i % ld4);
    const uint32_t id = sorted_ids[p];
    const uint32_t c = assign[id];
    if (owner != nullptr && owner[c] != rank) continue;  // another GPU's list
    const uint64_t dst = (uint64_t)list_off[c] + (p - starts[c]);
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (c4 < ldx4 && c4 * 4 < d) {
      v = reinterpret_cast<const f32x4*>(X + (uint64_t)id * ldx)[c4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c4 * 4 + u >= d) v[u] = 0.0f;
    }
    *reinterpret_cast<f32x4*>(rows + blocked_index(dst, c4 * 4, ld)) = v;
    if (c4 == 0) row_ids[dst] = id;
  }
}

// The same placement, one BLOCK per destination tile of 64 storage rows: the 64 source rows are read as they lie (3 KB
// contiguous each), turned through LDS 64 float4 columns at a time, and written as the tile's contiguous 1 KiB pieces.
// (gather_rows_kernel above writes every float4 to its own piece: 16 useful bytes per 64-byte sector and a stride of 1 KiB
// between consecutive threads -- 69 ms for N = 10M x 768, 0.9 TB/s, the largest serial-looking kernel of build_index.)
// tile_list[t] = the list tile t belongs to (lists start on tile boundaries); rows of the tile past the list's length are
// written as zeros (slack for `add`), their row_ids stay 0xFFFFFFFF.
constexpr uint32_t kGatherCols4 = 64;  // float4 columns per pass: 64 rows x 65 float4 = 66.5 KB of LDS
// row_src != nullptr (the receive side of the row-sharded build): storage row r holds source row row_src[r] of X (0xFFFFFFFF: none)
// and its vec id is src_ids[that row]; otherwise the source of a row follows from the cluster-sorted order (sorted_ids / starts).
__global__ __launch_bounds__(256) void gather_tiles_kernel(const float* X, uint32_t ldx, uint32_t d, uint32_t ld, const uint32_t* sorted_ids,
                                                           const uint32_t* starts, const uint32_t* list_off, const uint32_t* list_len,
                                                           const uint32_t* tile_list, float* rows, uint32_t* row_ids,
                                                           const uint32_t* row_src = nullptr, const uint32_t* src_ids = nullptr) {
  extern __shared__ __attribute__((aligned(16))) f32x4 tl[];  // [64][kGatherCols4 + 1]
  __shared__ uint32_t s_id[kWave];
  const uint32_t t = blockIdx.x, c = tile_list[t];
  const uint32_t row0 = t * 64u, in_list0 = row0 - list_off[c], len = list_len[c];
  const uint32_t n_valid = in_list0 < len ? (len - in_list0 < 64u ? len - in_list0 : 64u) : 0u;
  if (threadIdx.x < 64) {
    uint32_t id = 0xFFFFFFFFu;
    if (threadIdx.x < n_valid) id = row_src ? row_src[row0 + threadIdx.x] : sorted_ids[starts[c] + in_list0 + threadIdx.x];
    s_id[threadIdx.x] = id;
    if (id != 0xFFFFFFFFu) row_ids[row0 + threadIdx.x] = src_ids ? src_ids[id] : id;
  }
  __syncthreads();
  const uint32_t ld4 = ld / 4, ldx4 = ldx / 4;
  f32x4* tile = reinterpret_cast<f32x4*>(rows + (uint64_t)t * 64ull * ld);
  constexpr uint32_t kPitch = kGatherCols4 + 1;
  for (uint32_t c0 = 0; c0 < ld4; c0 += kGatherCols4) {
    const uint32_t nc = ld4 - c0 < kGatherCols4 ? ld4 - c0 : kGatherCols4;
    for (uint32_t i = threadIdx.x; i < 64u * kGatherCols4; i += 256u) {  // a row's float4s by consecutive threads
      const uint32_t r = i / kGatherCols4, j = i % kGatherCols4, c4 = c0 + j;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      const uint32_t id = s_id[r];
      if (j < nc && id != 0xFFFFFFFFu && c4 < ldx4 && c4 * 4 < d) {
        v = reinterpret_cast<const f32x4*>(X + (uint64_t)id * ldx)[c4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (c4 * 4 + u >= d) v[u] = 0.0f;  // (columns >= d of X are the caller's padding and may hold anything)
      }
      tl[r * kPitch + j] = v;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 64u * nc; i += 256u) {  // a piece's 64 rows by consecutive threads: 1 KiB contiguous
      const uint32_t j = i / 64u, r = i % 64u;
      tile[(uint64_t)(c0 + j) * 64 + r] = tl[r * kPitch + j];
    }
    __syncthreads();
  }
}

// one padded row (ld floats, row-major) -> storage row `dst` of the blocked matrix
__global__ void scatter_row_kernel(const float* row, uint32_t ld, uint64_t dst, float* rows) {
  const uint32_t c4 = blockIdx.x * blockDim.x + threadIdx.x;
  if (c4 < ld / 4)
    *reinterpret_cast<f32x4*>(rows + blocked_index(dst, c4 * 4, ld)) = reinterpret_cast<const f32x4*>(row)[c4];
}

__global__ void gather_init_kernel(const float* X, uint32_t ldx, uint32_t d, uint32_t ldc, const uint32_t* idx, uint32_t k, float* C) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)k * ldc) return;
  const uint32_t j = (uint32_t)(i % ldc);
  C[i] = j < d ? X[(uint64_t)idx[i / ldc] * ldx + j] : 0.0f;
}

__global__ void u32_to_u64_kernel(const uint32_t* in, uint64_t n, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}

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
  static constexpr uint32_t kEvRing = 64;  // scan-launch timing ring (measurement hook)
  hipEvent_t ev0[kEvRing] = {}, ev1[kEvRing] = {};
  hipEvent_t evc[3] = {};  // batched coarse quantiser of the most recent search: before the GEMM | after it | after select / re-score
  bool evc_valid = false;
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
static thread_local SearchWs* W = nullptr;  // the workspace leased by the call running on this thread

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
  // matrix-core list scan (prescan.hip.h): |x|^2 per storage row, [0] max |x|^2 bits, [1] certificate failures (running)
  DevBuf xnorm, pre_misc;
  // fp16 shadow of the rows for the matrix-core pre-selection (+50 % corpus memory; VERS_SHADOW=0 or a failed
  // allocation: the f32 rows feed it).  The wider certificate window makes it sensitive to data with many near-ties:
  // the failure counter is watched through a pinned word and the shadow is switched off for the handle when more
  // than 1/8 of the queries had to be re-scanned exactly.
  DevBuf rows_bf;
  // Row-major second copy of the stored rows for the exact finish: a candidate row of the lane-transposed tile layout is
  // 192 separate 16-byte pieces (one per 64-byte sector: 4x the useful bytes, 74 us per batch of 1024 at cfg3 whatever the
  // shard count); row-major it is 3 KB of whole sectors.  OPT-IN (VERS_ROWMAJOR=1; -1 = whenever the rows take at most a
  // quarter of the device's memory): measured at cfg3, same box -- exact finish 74 -> 53 us, but the list scan 10 us
  // slower with twice the rows mapped, net -8 us per step at 8 ranks and nothing on one GPU: not worth doubling the
  // corpus memory by default.  Same bits either way.
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

namespace {

int32_t status_to_rc(vers_ivf* h, uint32_t s, uint32_t slot);

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
  if (s & kStNaN) return fail(VERS_ERR_NAN, "NaN distance (the reference panics in partial_cmp().unwrap())");
  if (s & kStInsufficient)
    return fail(VERS_ERR_INSUFFICIENT, "fewer than top_k vectors reachable (reference: index out of bounds, ivfflat.rs:169)");
  if (s & kStSpillTooDeep) {
    fail(VERS_ERR_INVALID, "search_approximate spills past the lists this device-pointer call ranked (48): the host-pointer entry point retries with every list");
    return kRetrySpill;
  }
  return VERS_OK;
}
// host-pointer calls and add run with status word 1 while they hold the handle's mutex
struct HostStatusSlot {
  SearchWs* w;
  explicit HostStatusSlot(vers_ivf*) : w(W) { w->st_slot = 1; }
  ~HostStatusSlot() { w->st_slot = 0; }
};
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

// test hook: every storage row that holds no vector (slack behind the lists, tile padding) gets `value` in all its columns,
// then the derived arrays (|x|^2, fp16 shadow, residual) are rebuilt -- what uninitialised device memory may look like
static __global__ void poison_slack_kernel(float* rows, uint32_t ld, const uint32_t* row_ids, uint64_t n_rows, float value) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t r = t / (ld / 4);
  const uint32_t j = (uint32_t)(t % (ld / 4));
  if (r >= n_rows || row_ids[r] != 0xFFFFFFFFu) return;
  reinterpret_cast<f32x4*>(rows + (r >> 6) * 64ull * ld)[(uint64_t)j * 64 + (r & 63)] = f32x4{value, value, value, value};
}
// vers_set_option("scan_events", v): HIP event records around every list-scan launch (vers_ivf_last_scan / vers_ivf_scan_times).
// 1 always, 0 never, 2 (default) for batches only: the two records cost a single-query call 5.5-6 us of ~100 (same-box A/B,
// scripts/bench_host_b1.py), a batch of 1024 nothing measurable.
inline std::atomic<int>& scan_events_ref() {
  static std::atomic<int> m{[] { const char* e = getenv("VERS_SCAN_EVENTS"); return e ? atoi(e) : 2; }()};
  return m;
}
inline std::atomic<int>& shadow_mode_ref() {  // VERS_SHADOW (default 1) / vers_set_option("shadow", v)
  static std::atomic<int> m{[] { const char* e = getenv("VERS_SHADOW"); return e ? (atoi(e) != 0 ? 1 : 0) : 1; }()};
  return m;
}
inline int shadow_mode() { return shadow_mode_ref().load(std::memory_order_relaxed); }
// |x|^2 of storage rows [r_begin, r_end) for the matrix-core list scan; a full refresh also resets the maximum
int32_t refresh_norms(vers_ivf* h, uint64_t r_begin, uint64_t r_end, hipStream_t st) {
  if (int32_t rc = h->pre_misc.reserve(64)) return rc;
  const bool full = r_begin == 0 && r_end == h->cap_rows;
  if (full) {
    if (int32_t rc = h->xnorm.reserve((h->cap_rows ? h->cap_rows : 1) * sizeof(float))) return rc;
    VERS_HIP_TRY(hipMemsetAsync(h->pre_misc.p, 0, 64, st));
    for (auto& w : h->pool)  // a new index: ranked lists computed ahead belong to the old centroids (the caller holds the handle exclusively)
      for (auto& a : w->ahead) a.valid = false;
  }
  // fp16 shadow of the rows for the matrix-core list scan of batches (prescan.hip.h): on unless VERS_SHADOW=0 /
  // vers_set_option("shadow", 0) at build / upload time.
  const bool shadow = shadow_mode() != 0;
  if (full) h->shadow_valid = false;
  if (shadow) {
    if (full) {
      const size_t need = (h->cap_rows ? h->cap_rows : 1) * (size_t)h->ld * sizeof(uint16_t);
      if (need > h->rows_bf.cap) {  // optional memory: without it (or with less than 4 GB left for the searches' scratch) the f32 rows stay in charge
        h->rows_bf.release();
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        void* pbf = nullptr;
        if (need + (size_t(4) << 30) <= free_b && hipMalloc(&pbf, need) == hipSuccess) { h->rows_bf.p = pbf; h->rows_bf.cap = need; dev_mem_account((int64_t)need); }
        else (void)hipGetLastError();
      }
      h->shadow_valid = h->rows_bf.p != nullptr && h->rows_bf.cap >= need;
      h->shadow_off = false; h->shadow_queries = 0;  // (the failure counter in pre_misc was just zeroed)
      if (!h->fail_watch) VERS_HIP_TRY(hipHostMalloc((void**)&h->fail_watch, 64, hipHostMallocDefault));
      *h->fail_watch = 0;
    }
    if (r_end > r_begin && h->shadow_valid) {
      const uint64_t work = (r_end - r_begin) * (h->ld / 8);
      hipLaunchKernelGGL(rows_to_f16_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, h->rows.as<float>(), h->ld, r_begin, r_end,
                         h->rows_bf.as<uint16_t>());
      hipLaunchKernelGGL(shadow_residual_kernel, dim3((unsigned)((r_end - r_begin + 255) / 256)), dim3(256), 0, st, h->rows.as<float>(), h->ld,
                         h->row_ids.as<uint32_t>(), r_begin, r_end, h->pre_misc.as<uint32_t>() + 2);
      VERS_HIP_TRY(hipGetLastError());
    }
  } else {
    h->shadow_valid = false;  // rows changed without their shadow following
    if (full) h->rows_bf.release();
  }
  {
    static const int rm_mode = [] { const char* e = getenv("VERS_ROWMAJOR"); return e ? atoi(e) : 0; }();  // opt-in: see vers_ivf::rows_rm
    if (full) {
      size_t free_b = 0, total_b = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      const size_t need = (h->cap_rows ? h->cap_rows : 1) * (size_t)h->ld * sizeof(float);
      const bool want = rm_mode == 1 || (rm_mode < 0 && need <= total_b / 4);
      if (!want || need > h->rows_rm.cap) h->rows_rm.release();
      if (want && h->rows_rm.p == nullptr) {  // optional memory: a failed allocation leaves the tile gather in charge
        void* prm = nullptr;
        if (need + (size_t(2) << 30) <= free_b && hipMalloc(&prm, need) == hipSuccess) { h->rows_rm.p = prm; h->rows_rm.cap = need; dev_mem_account((int64_t)need); }
        else (void)hipGetLastError();
      }
    }
    if (r_end > r_begin && h->rows_rm.p)
      if (int32_t rc = launch_from_blocked(h->rows.as<float>(), h->ld, r_begin, r_end - r_begin, h->ld, h->rows_rm.as<float>() + r_begin * (size_t)h->ld,
                                           h->ld, st))
        return rc;
  }
  if (r_end > r_begin) {
    hipLaunchKernelGGL(blocked_row_norms_kernel, dim3((unsigned)((r_end - r_begin + 255) / 256)), dim3(256), 0, st, h->rows.as<float>(), h->ld,
                       h->row_ids.as<uint32_t>(), r_begin, r_end, h->xnorm.as<float>(), h->pre_misc.as<uint32_t>());
    VERS_HIP_TRY(hipGetLastError());
  }
  return VERS_OK;
}

// ---- build: storage layout, row placement, k-means ------------------------------------------------------------
// How the rows of a build are spread over processes.  comm == nullptr: one process holds all n rows.
struct BuildShard {
  const vers_comm_t* comm = nullptr;
  uint32_t rank = 0, world = 1;
  uint64_t row_begin = 0;  // global index of this process's first row
  uint64_t n_total = 0;    // rows over all processes
  const struct Agreement* agree = nullptr;  // (multi-process builds: made by build_common before the first collective)
};

int32_t comm_rc(int32_t rc, const char* what) {
  if (rc) return fail(VERS_ERR_COMM, std::string("vers_comm_t::") + what + " reported failure (status " + std::to_string(rc) + ")");
  return VERS_OK;
}

// Failure propagation of the row-sharded build: every callback is a rendezvous, so a rank that returned early (a failed
// allocation, a HIP error in its assign pass) would leave its peers blocked inside the next one -- under RCCL a spinning
// kernel until the watchdog fires.  At the points where a rank can fail on its own, right before the ranks next meet, all
// ranks exchange how they fared (one 4-byte all_gather) and LEAVE TOGETHER when anyone failed.  (What cannot be agreed on
// is a failure of the communicator itself: the host must abort the process group when any rank returns non-zero.)
// The 4 + 4 W bytes it needs are allocated ONCE per build, before the first collective (Agreement::init: a failure there is
// returned before any rank has entered a rendezvous -- the host aborts the group as for any non-zero return): agree() itself
// never allocates, and it ALWAYS enters the all_gather -- with its error code when the local staging copy failed -- so that
// an out-of-memory rank, the very situation it exists for, cannot strand its peers inside the collective.
struct Agreement {
  const vers_comm_t* cm = nullptr;
  uint32_t W = 1;
  DevBuf mine, all;
  int32_t init(const vers_comm_t* comm, uint32_t world) {
    cm = comm; W = world;
    if (cm == nullptr || W <= 1) return VERS_OK;
    if (int32_t rc = mine.reserve(16)) return rc;
    if (int32_t rc = all.reserve(16 * (size_t)W)) return rc;
    VERS_HIP_TRY(hipMemset(mine.p, 0, 16));
    return VERS_OK;
  }
  int32_t operator()(int32_t my_rc, const char* where) const {
    if (cm == nullptr || W <= 1) return my_rc;
    int32_t word = my_rc;
    bool staged = hipMemcpy(mine.p, &word, 4, hipMemcpyHostToDevice) == hipSuccess;
    if (!staged) {  // (the device is in trouble: say so with whatever still works, then meet the peers all the same)
      (void)hipGetLastError();
      staged = hipMemset(mine.p, 0xFF, 4) == hipSuccess;
      if (!staged) (void)hipGetLastError();
      if (!my_rc) my_rc = fail(VERS_ERR_HIP, "hipMemcpy failed while staging the agreement word");
    }
    if (int32_t rc = comm_rc(cm->all_gather(cm->ctx, mine.p, all.p, 4), "all_gather")) return my_rc ? my_rc : rc;
    std::vector<int32_t> got(W, 0);
    if (hipMemcpy(got.data(), all.p, 4 * (size_t)W, hipMemcpyDeviceToHost) != hipSuccess) return my_rc ? my_rc : fail(VERS_ERR_HIP, "hipMemcpy failed");
    if (my_rc) return my_rc;
    for (uint32_t r = 0; r < W; ++r)
      if (got[r]) return fail(VERS_ERR_COMM, std::string("rank ") + std::to_string(r) + " failed in " + where + " (status " + std::to_string(got[r]) + "): every rank leaves the build");
    return VERS_OK;
  }
};

// Storage plan from the GLOBAL list lengths: owners (LPT when sharded), offsets and capacities of the owned lists,
// device tables, zeroed row ids.
int32_t plan_storage(vers_ivf* h, const uint32_t* lens, uint32_t k, hipStream_t st) {
  h->h_len.assign(lens, lens + k);
  h->h_off.assign(k, 0);
  h->h_cap.assign(k, 0);
  uint64_t off = 0;
  h->max_len = 0;
  h->h_owner.assign(k, 0);
  if (h->world > 1) {
    std::vector<uint64_t> l64(h->h_len.begin(), h->h_len.end());
    shard_plan(l64.data(), k, h->world, h->h_owner.data());
  }
  for (uint32_t c = 0; c < k; ++c) {
    const uint32_t len = h->h_len[c];
    const bool mine = h->h_owner[c] == h->rank;
    const uint32_t cap = mine ? round_up(len + std::max<uint32_t>(64u, len / 16u), 64u) : 0u;
    h->h_off[c] = (uint32_t)off;
    h->h_cap[c] = cap;
    off += cap;
    h->max_len = std::max(h->max_len, len);
    if (off > 0xFFFFFFFFull) return fail(VERS_ERR_INVALID, "more than 2^32-1 storage rows on one GPU");
  }
  if (int32_t rc = h->owner.reserve(k ? k : 1)) return rc;
  if (k) VERS_HIP_TRY(hipMemcpyAsync(h->owner.p, h->h_owner.data(), k, hipMemcpyHostToDevice, st));
  {  // longest lists first (stable: ties by index); add() changes lengths by one at a time, the order is kept as it is
    std::vector<uint32_t> ord(k), so(k ? k : 1), sl(k ? k : 1);
    for (uint32_t c = 0; c < k; ++c) ord[c] = c;
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t x, uint32_t y) { return h->h_len[x] > h->h_len[y]; });
    h->h_slot.assign(k, 0);
    for (uint32_t i = 0; i < k; ++i) { h->h_slot[ord[i]] = i; so[i] = h->h_off[ord[i]]; sl[i] = h->h_len[ord[i]]; }
    if (int32_t rc = h->list_slot.reserve((k ? k : 1) * sizeof(uint32_t))) return rc;
    if (int32_t rc = h->slot_off.reserve((k ? k : 1) * sizeof(uint32_t))) return rc;
    if (int32_t rc = h->slot_len.reserve((k ? k : 1) * sizeof(uint32_t))) return rc;
    if (k) {
      VERS_HIP_TRY(hipMemcpy(h->list_slot.p, h->h_slot.data(), (size_t)k * 4, hipMemcpyHostToDevice));
      VERS_HIP_TRY(hipMemcpy(h->slot_off.p, so.data(), (size_t)k * 4, hipMemcpyHostToDevice));
      VERS_HIP_TRY(hipMemcpy(h->slot_len.p, sl.data(), (size_t)k * 4, hipMemcpyHostToDevice));
    }
  }
  {
    std::vector<uint32_t> asc(h->h_len);
    std::sort(asc.begin(), asc.end());
    h->len_asc_prefix.assign(k, 0);
    uint64_t run = 0;
    for (uint32_t i = 0; i < k; ++i) { run += asc[i]; h->len_asc_prefix[i] = run; }
  }
  h->cap_rows = off;
  {
    std::vector<uint32_t> tl((size_t)(off / 64) ? (size_t)(off / 64) : 1, 0u);
    for (uint32_t c = 0; c < k; ++c)
      for (uint32_t r = 0; r < h->h_cap[c]; r += 64) tl[(h->h_off[c] + r) / 64] = c;
    if (int32_t rc = h->tile_list.reserve(tl.size() * sizeof(uint32_t))) return rc;
    VERS_HIP_TRY(hipMemcpy(h->tile_list.p, tl.data(), tl.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  if (int32_t rc = h->rows.reserve((off ? off : 1) * (size_t)h->ld * sizeof(float))) return rc;
  if (int32_t rc = h->row_ids.reserve((off ? off : 1) * sizeof(uint32_t))) return rc;
  if (int32_t rc = h->list_off.reserve((k ? k : 1) * sizeof(uint32_t))) return rc;
  if (int32_t rc = h->list_len.reserve((k ? k : 1) * sizeof(uint32_t))) return rc;
  VERS_HIP_TRY(hipMemsetAsync(h->row_ids.p, 0xFF, (off ? off : 1) * sizeof(uint32_t), st));
  if (k) {
    VERS_HIP_TRY(hipMemcpyAsync(h->list_off.p, h->h_off.data(), (size_t)k * 4, hipMemcpyHostToDevice, st));
    VERS_HIP_TRY(hipMemcpyAsync(h->list_len.p, h->h_len.data(), (size_t)k * 4, hipMemcpyHostToDevice, st));
  }
  return VERS_OK;
}

// Everything of the index that derives from h->centroids and the stored rows: centroids in the scan layout and as
// MFMA operands, |c|^2, |x|^2.  The index is complete (and the stream idle) on return.
int32_t finish_index(vers_ivf* h, uint32_t k, uint64_t n_total, hipStream_t st) {
  if (int32_t rc = h->centroids_b.reserve(std::max<uint64_t>(1, blocked_floats(k, h->ld)) * sizeof(float))) return rc;
  if (int32_t rc = launch_to_blocked(h->centroids.as<float>(), h->ldx, h->d, k, h->centroids_b.as<float>(), h->ld, st)) return rc;
  h->k_pad = round_up(k ? k : 1, kGemmBN);
  if (int32_t rc = h->centroids_g.reserve((size_t)h->k_pad * h->ldq * sizeof(float))) return rc;
  if (int32_t rc = h->cnorm.reserve((size_t)h->k_pad * sizeof(float))) return rc;
  if (int32_t rc = h->coarse_stat.reserve(16)) return rc;
  VERS_HIP_TRY(hipMemsetAsync(h->centroids_g.p, 0, (size_t)h->k_pad * h->ldq * sizeof(float), st));
  VERS_HIP_TRY(hipMemsetAsync(h->coarse_stat.p, 0, 16, st));
  if (int32_t rc = launch_stage_queries(h->centroids.as<float>(), h->ldx, h->d, h->centroids_g.as<float>(), h->ldq, k, 1, st)) return rc;
  hipLaunchKernelGGL(row_norms_kernel, dim3((h->k_pad + 255) / 256), dim3(256), 0, st, h->centroids_g.as<float>(), h->ldq, k, h->k_pad,
                     h->cnorm.as<float>());
  VERS_HIP_TRY(hipGetLastError());
  {  // bf16 hi | lo halves of the same matrix: the N operand of the batched coarse quantiser's bf16x3 contraction
    const size_t ne = (size_t)h->k_pad * h->ldq;
    if (int32_t rc = h->centroids_gs.reserve(2 * ne * sizeof(uint16_t))) return rc;
    VERS_HIP_TRY(launch_split_bf16(h->centroids_g.as<float>(), ne, h->centroids_gs.as<__bf16>(), h->centroids_gs.as<__bf16>() + ne, st));
  }
  std::vector<float> cn(k ? k : 1, 0.0f);
  if (k) VERS_HIP_TRY(hipMemcpyAsync(cn.data(), h->cnorm.p, (size_t)k * sizeof(float), hipMemcpyDeviceToHost, st));
  VERS_HIP_TRY(hipStreamSynchronize(st));
  h->cmax2 = 0.0f;
  for (uint32_t c = 0; c < k; ++c) h->cmax2 = std::max(h->cmax2, cn[c]);  // NaN centroids never raise it; they fail the certificate
  h->k = k;
  h->n_total = n_total;
  {  // diagnosis (VERS_POISON_SLACK=inf|nan|<number>): every new index starts with that value in the rows that hold no vector
    static const char* poison = getenv("VERS_POISON_SLACK");
    if (poison && h->cap_rows) {
      const float v = (float)atof(poison);  // ("inf" and "nan" parse as such)
      const uint64_t work = h->cap_rows * (h->ld / 4);
      hipLaunchKernelGGL(poison_slack_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, h->rows.as<float>(), h->ld,
                         (const uint32_t*)h->row_ids.as<uint32_t>(), h->cap_rows, v);
      VERS_HIP_TRY(hipGetLastError());
    }
  }
  if (int32_t rc = refresh_norms(h, 0, h->cap_rows, st)) return rc;
  // |x|^2, max |x|^2 and the optional shadow were queued on `st`; searches run on other (possibly non-blocking)
  // streams and a certificate evaluated against a stale maximum would be unsound: the index is complete on return
  VERS_HIP_TRY(hipStreamSynchronize(st));
  return VERS_OK;
}

// index from (X in vec_id order -- ALL rows in this process --, centroids already in h->centroids, device assignments);
// with vers_ivf_set_shard only the owned lists are stored.
int32_t install_index(vers_ivf* h, const float* X, uint32_t ldx, uint64_t n, const uint32_t* d_assign, uint32_t k,
                      hipStream_t st) {
  DevBuf sorted;
  if (int32_t rc = sorted.reserve((n ? n : 1) * sizeof(uint32_t))) return rc;
  if (int32_t rc = h->km.counts.reserve((2 * (size_t)k + 2) * sizeof(uint32_t))) return rc;
  uint32_t* counts = h->km.counts.as<uint32_t>();
  uint32_t* starts = counts + k;
  if (int32_t rc = km_group(d_assign, (uint32_t)n, k, sorted.as<uint32_t>(), counts, starts, h->km, st)) return rc;
  std::vector<uint32_t> lens(k ? k : 1, 0);
  if (k) VERS_HIP_TRY(hipMemcpyAsync(lens.data(), counts, (size_t)k * 4, hipMemcpyDeviceToHost, st));
  VERS_HIP_TRY(hipStreamSynchronize(st));
  if (int32_t rc = plan_storage(h, lens.data(), k, st)) return rc;
  if (n) {
    static const bool by_tile = [] { const char* e = getenv("VERS_GATHER_TILES"); return !e || atoi(e) != 0; }();
    if (by_tile && h->cap_rows >= 64) {
      const size_t lds = 64 * (size_t)(kGatherCols4 + 1) * sizeof(f32x4);
      if (int32_t rc = scan_prepare_launch(gather_tiles_kernel, lds)) return rc;
      hipLaunchKernelGGL(gather_tiles_kernel, dim3((unsigned)(h->cap_rows / 64)), dim3(256), lds, st, X, ldx, h->d, h->ld, sorted.as<uint32_t>(),
                         (const uint32_t*)starts, h->list_off.as<uint32_t>(), h->list_len.as<uint32_t>(), h->tile_list.as<uint32_t>(),
                         h->rows.as<float>(), h->row_ids.as<uint32_t>());
    } else
    hipLaunchKernelGGL(gather_rows_kernel, dim3(h->n_cu * 8), dim3(256), 0, st, X, ldx, h->d, h->ld, sorted.as<uint32_t>(), d_assign, starts,
                       h->list_off.as<uint32_t>(), h->world > 1 ? h->owner.as<uint8_t>() : (const uint8_t*)nullptr, h->rank, n,
                       h->rows.as<float>(), h->row_ids.as<uint32_t>());
    VERS_HIP_TRY(hipGetLastError());
  }
  return finish_index(h, k, n, st);
}

// One destination segment of the row exchange: `count` consecutive rows of the receive buffer (one source rank's
// members of one owned list, ascending vec_id) go to storage rows dest, dest + 1, ...
struct RecvSeg {
  uint32_t src_row, count, dest, pad;
};

// send side: local rows in (destination rank, cluster, ascending index) order, row-major pitch ldp, + their vec ids
__global__ void pack_rows_kernel(const float* X, uint32_t ldx, uint32_t d, uint32_t ldp, const uint32_t* sorted_ids, const uint32_t* assign,
                                 const uint32_t* starts, const uint32_t* send_base, uint32_t row_begin, uint64_t n, float* out, uint32_t* out_ids) {
  const uint32_t ldp4 = ldp / 4, ldx4 = ldx / 4;
  const uint64_t total = n * ldp4;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t p = i / ldp4;
    const uint32_t c4 = (uint32_t)(i % ldp4);
    const uint32_t id = sorted_ids[p];
    const uint32_t c = assign[id];
    const uint64_t dst = (uint64_t)send_base[c] + (p - starts[c]);
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (c4 < ldx4 && c4 * 4 < d) {
      v = reinterpret_cast<const f32x4*>(X + (uint64_t)id * ldx)[c4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c4 * 4 + u >= d) v[u] = 0.0f;
    }
    reinterpret_cast<f32x4*>(out + dst * ldp)[c4] = v;
    if (c4 == 0) out_ids[dst] = row_begin + id;
  }
}

// receive side: block per segment, rows into the lane-transposed tiles of their list
__global__ __launch_bounds__(256) void unpack_rows_kernel(const float* in, uint32_t ldp, const uint32_t* in_ids, const RecvSeg* segs, uint32_t ld,
                                                          float* rows, uint32_t* row_ids) {
  const RecvSeg sg = segs[blockIdx.x];
  const uint32_t ld4 = ld / 4, ldp4 = ldp / 4;
  const uint64_t total = (uint64_t)sg.count * ld4;
  for (uint64_t i = threadIdx.x; i < total; i += blockDim.x) {
    const uint32_t r = (uint32_t)(i / ld4), c4 = (uint32_t)(i % ld4);
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (c4 < ldp4) v = reinterpret_cast<const f32x4*>(in + (uint64_t)(sg.src_row + r) * ldp)[c4];
    *reinterpret_cast<f32x4*>(rows + blocked_index((uint64_t)sg.dest + r, c4 * 4, ld)) = v;
    if (c4 == 0) row_ids[sg.dest + r] = in_ids[sg.src_row + r];
  }
}

// storage row -> row of the receive buffer, from the segments (a block per segment)
__global__ void fill_row_src_kernel(const RecvSeg* segs, uint32_t* row_src) {
  const RecvSeg sg = segs[blockIdx.x];
  for (uint32_t r = threadIdx.x; r < sg.count; r += blockDim.x) row_src[sg.dest + r] = sg.src_row + r;
}

__global__ void sum_counts_kernel(const uint32_t* counts_all, uint32_t world, uint32_t k, uint32_t* out) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= k) return;
  uint32_t s = 0;
  for (uint32_t r = 0; r < world; ++r) s += counts_all[(uint64_t)r * k + c];
  out[c] = s;
}

__global__ void scatter_centroid_rows_kernel(const float* tmp, uint32_t ld, const uint32_t* dst_c, uint32_t cnt, float* C) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint64_t)cnt * ld) return;
  C[(uint64_t)dst_c[i / ld] * ld + i % ld] = tmp[i];
}

// Row-sharded install (ivfflat.rs:123-127 across processes): the lists are dealt to the ranks by LPT over the GLOBAL
// lengths and every rank ships each of its rows to the owner of the row's list with ONE all_to_all_v (rows) + one for
// the vec ids.  A rank sends its rows ordered by (destination, cluster, ascending local index); ranks hold ascending
// ranges, so concatenating the sources in rank order inside a list IS the reference's ascending vec_id order.
int32_t install_index_sharded(vers_ivf* h, const float* X, uint32_t ldx, uint64_t n_loc, const BuildShard& sh, const uint32_t* d_assign,
                              uint32_t k, hipStream_t st) {
  const vers_comm_t* cm = sh.comm;
  const uint32_t W = sh.world, me = sh.rank;
  DevBuf sorted, counts_all_d;
  uint32_t* counts = nullptr;
  uint32_t* starts = nullptr;
  const int32_t rc_group = [&]() -> int32_t {  // (local work ahead of the first rendezvous of the install: agreed on before anyone enters it)
    if (int32_t rc = sorted.reserve((n_loc ? n_loc : 1) * sizeof(uint32_t))) return rc;
    if (int32_t rc = h->km.counts.reserve((2 * (size_t)k + 2) * sizeof(uint32_t))) return rc;
    counts = h->km.counts.as<uint32_t>();
    starts = counts + k;
    if (int32_t rc = km_group(d_assign, (uint32_t)n_loc, k, sorted.as<uint32_t>(), counts, starts, h->km, st)) return rc;
    if (int32_t rc = counts_all_d.reserve((size_t)W * (k ? k : 1) * 4)) return rc;
    VERS_HIP_TRY(hipStreamSynchronize(st));
    return VERS_OK;
  }();
  if (int32_t rc = sh.agree ? (*sh.agree)(rc_group, "grouping the local rows by list") : rc_group) return rc;
  if (k)
    if (int32_t rc = comm_rc(cm->all_gather(cm->ctx, counts, counts_all_d.p, (uint64_t)k * 4), "all_gather")) return rc;
  std::vector<uint32_t> ca((size_t)W * (k ? k : 1), 0), starts_h((size_t)k + 1, 0);
  if (k) {
    VERS_HIP_TRY(hipMemcpy(ca.data(), counts_all_d.p, (size_t)W * k * 4, hipMemcpyDeviceToHost));
    VERS_HIP_TRY(hipMemcpy(starts_h.data(), starts, ((size_t)k + 1) * 4, hipMemcpyDeviceToHost));
  }
  std::vector<uint32_t> lens(k ? k : 1, 0);
  for (uint32_t c = 0; c < k; ++c) {
    uint64_t s = 0;
    for (uint32_t r = 0; r < W; ++r) s += ca[(size_t)r * k + c];
    if (s > 0xFFFFFFFFull) return fail(VERS_ERR_INVALID, "a list longer than 2^32-1 rows");
    lens[c] = (uint32_t)s;
  }
  h->rank = me;
  h->world = W;
  const int32_t rc_plan = plan_storage(h, lens.data(), k, st);  // (a failure here -- the rows of the owned lists do not fit -- is agreed on below, with the exchange buffers)
  // send plan: rows for destination t = my members of the lists t owns, clusters ascending
  const uint32_t ldp = h->ldx;  // packed rows travel with the k-means pitch (d rounded up to 4 floats)
  std::vector<uint64_t> send_rows(W, 0), send_off_rows(W, 0), recv_rows(W, 0), recv_off_rows(W, 0);
  for (uint32_t c = 0; c < k; ++c) send_rows[h->h_owner[c]] += ca[(size_t)me * k + c];
  for (uint32_t t = 1; t < W; ++t) send_off_rows[t] = send_off_rows[t - 1] + send_rows[t - 1];
  std::vector<uint32_t> send_base(k ? k : 1, 0);
  {
    std::vector<uint64_t> cur(send_off_rows);
    for (uint32_t c = 0; c < k; ++c) {
      send_base[c] = (uint32_t)cur[h->h_owner[c]];
      cur[h->h_owner[c]] += ca[(size_t)me * k + c];
    }
  }
  // receive plan: from source s my owned lists' members, clusters ascending
  std::vector<RecvSeg> segs;
  for (uint32_t s = 0; s < W; ++s) {
    for (uint32_t c = 0; c < k; ++c)
      if (h->h_owner[c] == me) recv_rows[s] += ca[(size_t)s * k + c];
    if (s) recv_off_rows[s] = recv_off_rows[s - 1] + recv_rows[s - 1];
  }
  {
    std::vector<uint32_t> before(k ? k : 1, 0);  // members of list c that came from ranks before s
    for (uint32_t s = 0; s < W; ++s) {
      uint64_t r0 = recv_off_rows[s];
      for (uint32_t c = 0; c < k; ++c) {
        if (h->h_owner[c] != me) continue;
        const uint32_t cnt = ca[(size_t)s * k + c];
        if (cnt) segs.push_back(RecvSeg{(uint32_t)r0, cnt, h->h_off[c] + before[c], 0u});
        r0 += cnt;
        before[c] += cnt;
      }
    }
  }
  const uint64_t n_recv = recv_off_rows[W - 1] + recv_rows[W - 1];
  DevBuf sbuf, sids, rbuf, rids, dbase, dsegs;
  const int32_t rc_alloc = [&]() -> int32_t {
    if (rc_plan) return rc_plan;
    if (n_recv > 0xFFFFFFFFull) return fail(VERS_ERR_INVALID, "more than 2^32-1 rows received by one rank");
    if (int32_t rc = sbuf.reserve((n_loc ? n_loc : 1) * (size_t)ldp * 4)) return rc;
    if (int32_t rc = sids.reserve((n_loc ? n_loc : 1) * 4)) return rc;
    if (int32_t rc = rbuf.reserve((n_recv ? n_recv : 1) * (size_t)ldp * 4)) return rc;
    if (int32_t rc = rids.reserve((n_recv ? n_recv : 1) * 4)) return rc;
    if (int32_t rc = dbase.reserve((k ? k : 1) * 4)) return rc;
    if (int32_t rc = dsegs.reserve((segs.size() ? segs.size() : 1) * sizeof(RecvSeg))) return rc;
    return VERS_OK;
  }();
  if (int32_t rc = sh.agree ? (*sh.agree)(rc_alloc, "the exchange buffers of the rows-to-owners all_to_all_v") : rc_alloc) return rc;  // (storage + send + receive: the build's peak)
  if (k) VERS_HIP_TRY(hipMemcpyAsync(dbase.p, send_base.data(), (size_t)k * 4, hipMemcpyHostToDevice, st));
  if (!segs.empty()) VERS_HIP_TRY(hipMemcpyAsync(dsegs.p, segs.data(), segs.size() * sizeof(RecvSeg), hipMemcpyHostToDevice, st));
  if (n_loc) {
    hipLaunchKernelGGL(pack_rows_kernel, dim3(h->n_cu * 8), dim3(256), 0, st, X, ldx, h->d, ldp, sorted.as<uint32_t>(), d_assign, starts,
                       dbase.as<uint32_t>(), (uint32_t)sh.row_begin, n_loc, sbuf.as<float>(), sids.as<uint32_t>());
    VERS_HIP_TRY(hipGetLastError());
  }
  VERS_HIP_TRY(hipStreamSynchronize(st));
  std::vector<uint64_t> sb(W), so(W), rb(W), ro(W);
  for (uint32_t t = 0; t < W; ++t) {
    sb[t] = send_rows[t] * ldp * 4; so[t] = send_off_rows[t] * ldp * 4;
    rb[t] = recv_rows[t] * ldp * 4; ro[t] = recv_off_rows[t] * ldp * 4;
  }
  if (int32_t rc = comm_rc(cm->all_to_all_v(cm->ctx, sbuf.p, sb.data(), so.data(), rbuf.p, rb.data(), ro.data()), "all_to_all_v")) return rc;
  for (uint32_t t = 0; t < W; ++t) {
    sb[t] = send_rows[t] * 4; so[t] = send_off_rows[t] * 4;
    rb[t] = recv_rows[t] * 4; ro[t] = recv_off_rows[t] * 4;
  }
  if (int32_t rc = comm_rc(cm->all_to_all_v(cm->ctx, sids.p, sb.data(), so.data(), rids.p, rb.data(), ro.data()), "all_to_all_v")) return rc;
  sbuf.release();
  sids.release();
  // (capacity slack and tile padding of the storage are zero rows: (0 - q)^2 terms never enter a result, ids stay 0xFFFFFFFF)
  static const bool by_tile = [] { const char* e = getenv("VERS_GATHER_TILES"); return !e || atoi(e) != 0; }();
  if (by_tile && h->cap_rows >= 64 && !segs.empty()) {
    // whole destination tiles through LDS (gather_tiles_kernel), every tile written completely: no memset of the storage
    DevBuf row_src;
    if (int32_t rc = row_src.reserve(h->cap_rows * sizeof(uint32_t))) return rc;
    VERS_HIP_TRY(hipMemsetAsync(row_src.p, 0xFF, h->cap_rows * sizeof(uint32_t), st));
    hipLaunchKernelGGL(fill_row_src_kernel, dim3((unsigned)segs.size()), dim3(256), 0, st, dsegs.as<RecvSeg>(), row_src.as<uint32_t>());
    const size_t lds = 64 * (size_t)(kGatherCols4 + 1) * sizeof(f32x4);
    if (int32_t rc = scan_prepare_launch(gather_tiles_kernel, lds)) return rc;
    hipLaunchKernelGGL(gather_tiles_kernel, dim3((unsigned)(h->cap_rows / 64)), dim3(256), lds, st, rbuf.as<float>(), ldp, h->d, h->ld,
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr, h->list_off.as<uint32_t>(), h->list_len.as<uint32_t>(),
                       h->tile_list.as<uint32_t>(), h->rows.as<float>(), h->row_ids.as<uint32_t>(), (const uint32_t*)row_src.as<uint32_t>(),
                       (const uint32_t*)rids.as<uint32_t>());
    VERS_HIP_TRY(hipGetLastError());
    VERS_HIP_TRY(hipStreamSynchronize(st));  // (row_src goes out of scope)
  } else {
    VERS_HIP_TRY(hipMemsetAsync(h->rows.p, 0, (h->cap_rows ? h->cap_rows : 1) * (size_t)h->ld * sizeof(float), st));
    if (!segs.empty()) {
      hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)segs.size()), dim3(256), 0, st, rbuf.as<float>(), ldp, rids.as<uint32_t>(),
                         dsegs.as<RecvSeg>(), h->ld, h->rows.as<float>(), h->row_ids.as<uint32_t>());
      VERS_HIP_TRY(hipGetLastError());
    }
  }
  VERS_HIP_TRY(hipStreamSynchronize(st));
  rbuf.release();
  rids.release();
  return finish_index(h, k, sh.n_total, st);
}

// build_kmeans + best-of-attempts (ivfflat.rs:73-121) on device-resident rows -- ALL of them (sh.comm == nullptr) or
// this process's contiguous range of a row-sharded corpus; leaves the winning centroids in h->centroids and the
// assignments of the LOCAL rows in best_assign.  Same arithmetic order either way (see vers_hip.h).
int32_t run_build(vers_ivf* h, const float* X, uint32_t ldx, uint64_t n, const BuildShard& sh, uint32_t k, uint64_t num_attempts,
                  uint64_t max_iterations, const uint64_t* init_indices, DevBuf& best_assign, float* out_cost, int32_t* out_kept,
                  uint64_t* out_iterations, hipStream_t st) {
  const uint32_t ld = h->ldx;  // centroids live row-major with pitch ldx during k-means
  const vers_comm_t* cm = sh.comm;
  const uint32_t W = sh.world, me = sh.rank;
  const bool multi = cm != nullptr && W > 1;
  // the ranges of all ranks (contiguous, ascending, covering 0 .. n_total)
  std::vector<uint64_t> begins(W + 1, 0);
  begins[W] = sh.n_total;
  if (multi) {
    DevBuf mine, all;
    if (int32_t rc = mine.reserve(16)) return rc;
    if (int32_t rc = all.reserve(16 * (size_t)W)) return rc;
    const uint64_t my[2] = {sh.row_begin, n};
    VERS_HIP_TRY(hipMemcpy(mine.p, my, 16, hipMemcpyHostToDevice));
    if (int32_t rc = comm_rc(cm->all_gather(cm->ctx, mine.p, all.p, 16), "all_gather")) return rc;
    std::vector<uint64_t> rg(2 * (size_t)W);
    VERS_HIP_TRY(hipMemcpy(rg.data(), all.p, 16 * (size_t)W, hipMemcpyDeviceToHost));
    uint64_t expect = 0;
    for (uint32_t r = 0; r < W; ++r) {
      if (rg[2 * r] != expect) return fail(VERS_ERR_INVALID, "vers_ivf_build_sharded_dev: the ranks' row ranges are not contiguous and ascending in rank order");
      begins[r] = rg[2 * r];
      expect += rg[2 * r + 1];
    }
    if (expect != sh.n_total) return fail(VERS_ERR_INVALID, "vers_ivf_build_sharded_dev: the ranks' row counts do not add up to n_total");
  }
  DevBuf C, Cn, S, assign, mind, sorted, idx, idx2, bestC, counts_all, counts_g, tmp_rows, ctl, ctl_all;
  const size_t cbytes = ((size_t)k * ld ? (size_t)k * ld : 1) * sizeof(float);
  const int32_t rc_alloc = [&]() -> int32_t {
    if (int32_t rc = C.reserve(cbytes)) return rc;
    if (int32_t rc = Cn.reserve(cbytes)) return rc;
    if (int32_t rc = bestC.reserve(cbytes)) return rc;
    if (int32_t rc = assign.reserve((n ? n : 1) * 4)) return rc;
    if (int32_t rc = best_assign.reserve((n ? n : 1) * 4)) return rc;
    if (int32_t rc = mind.reserve((n ? n : 1) * 4)) return rc;
    if (int32_t rc = sorted.reserve((n ? n : 1) * 4)) return rc;
    if (int32_t rc = idx.reserve((k ? k : 1) * 4)) return rc;
    if (int32_t rc = h->km.counts.reserve((2 * (size_t)k + 2) * 4)) return rc;
    if (int32_t rc = h->km.misc.reserve(64)) return rc;
    if (int32_t rc = h->km.status.reserve(16)) return rc;
    if (multi) {
      if (int32_t rc = S.reserve(cbytes)) return rc;
      if (int32_t rc = idx2.reserve((k ? k : 1) * 4)) return rc;
      if (int32_t rc = tmp_rows.reserve(cbytes)) return rc;
      if (int32_t rc = counts_all.reserve((size_t)W * (k ? k : 1) * 4)) return rc;
      if (int32_t rc = counts_g.reserve((k ? k : 1) * 4)) return rc;
      if (int32_t rc = ctl.reserve(16)) return rc;
      if (int32_t rc = ctl_all.reserve(16 * (size_t)W)) return rc;
    }
    return VERS_OK;
  }();
  auto agree = [&](int32_t rc, const char* where) -> int32_t { return multi && sh.agree ? (*sh.agree)(rc, where) : rc; };
  if (int32_t rc = agree(rc_alloc, "the build's allocations")) return rc;
  VERS_HIP_TRY(hipMemsetAsync(h->km.status.p, 0, 16, st));
  uint32_t* counts = h->km.counts.as<uint32_t>();
  uint32_t* starts = counts + k;
  float* cost_dev = h->km.misc.as<float>();
  float* cost_in = cost_dev + 1;
  uint32_t* flag_dev = h->km.misc.as<uint32_t>() + 4;
  float best = INFINITY;
  *out_kept = 0;
  const bool mfma = km_use_mfma(n, k, h->d);
  auto assign_pass = [&](const float* Cc, uint32_t* a_out, float* m_out) -> int32_t {
    if (n == 0) return VERS_OK;
    return (mfma ? km_assign_mfma : km_assign)(X, ldx, n, Cc, ld, k, h->d, a_out, m_out, h->km, h->n_cu, st, h->metric);
  };
  std::vector<uint32_t> src32(k ? k : 1), dst32(k ? k : 1);
  for (uint64_t a = 0; a < num_attempts; ++a) {
    if (sh.n_total > 0 && k == 0) return fail(VERS_ERR_EMPTY, "build_index with zero clusters: min_by over no centroids (reference panics)");
    for (uint32_t c = 0; c < k; ++c)
      if (init_indices[a * k + c] >= sh.n_total) return fail(VERS_ERR_INVALID, "vers_ivf_build: init index out of range");
    // initialize_centroids (ivfflat.rs:18-27, draws injected): C[c] = row init[c].  Sharded: the rows drawn from rank
    // r's range are gathered there and broadcast (bit copies), everyone scatters them to their centroid slots.
    for (uint32_t r = 0; r < W && k; ++r) {
      uint32_t cnt = 0;
      for (uint32_t c = 0; c < k; ++c) {
        const uint64_t ix = init_indices[a * k + c];
        if (ix >= begins[r] && ix < begins[r + 1]) { src32[cnt] = (uint32_t)(ix - begins[r]); dst32[cnt] = c; ++cnt; }
      }
      if (!multi) {  // one process: straight into C
        VERS_HIP_TRY(hipMemcpyAsync(idx.p, src32.data(), (size_t)k * 4, hipMemcpyHostToDevice, st));
        VERS_HIP_TRY(hipStreamSynchronize(st));  // src32 is reused by the next attempt
        hipLaunchKernelGGL(gather_init_kernel, dim3((unsigned)(((uint64_t)k * ld + 255) / 256)), dim3(256), 0, st, X, ldx, h->d, ld,
                           idx.as<uint32_t>(), k, C.as<float>());
        VERS_HIP_TRY(hipGetLastError());
        break;
      }
      if (cnt == 0) continue;
      if (r == me) {
        VERS_HIP_TRY(hipMemcpyAsync(idx.p, src32.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(gather_init_kernel, dim3((unsigned)(((uint64_t)cnt * ld + 255) / 256)), dim3(256), 0, st, X, ldx, h->d, ld,
                           idx.as<uint32_t>(), cnt, tmp_rows.as<float>());
        VERS_HIP_TRY(hipGetLastError());
      }
      VERS_HIP_TRY(hipMemcpyAsync(idx2.p, dst32.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, st));
      VERS_HIP_TRY(hipStreamSynchronize(st));
      if (int32_t rc = comm_rc(cm->broadcast(cm->ctx, tmp_rows.p, (uint64_t)cnt * ld * 4, r), "broadcast")) return rc;
      hipLaunchKernelGGL(scatter_centroid_rows_kernel, dim3((unsigned)(((uint64_t)cnt * ld + 255) / 256)), dim3(256), 0, st,
                         tmp_rows.as<float>(), ld, idx2.as<uint32_t>(), cnt, C.as<float>());
      VERS_HIP_TRY(hipGetLastError());
      VERS_HIP_TRY(hipStreamSynchronize(st));  // dst32 / tmp_rows are reused by the next source rank
    }
    uint64_t iters = 0;
    for (uint64_t it = 0; it < max_iterations; ++it) {
      {  // assign + grouping are local: what a rank's own failure there was is agreed on before the ranks next meet
        int32_t rc_a = assign_pass(C.as<float>(), assign.as<uint32_t>(), nullptr);
        if (!rc_a) rc_a = km_group(assign.as<uint32_t>(), (uint32_t)n, k, sorted.as<uint32_t>(), counts, starts, h->km, st);
        if (int32_t rc = agree(rc_a, "assign_to_clusters")) return rc;
      }
      if (!multi) {
        KmTimer t(st, &BuildStats::update_ms);
        if (int32_t rc = km_update(X, ldx, h->d, sorted.as<uint32_t>(), starts, counts, k, Cn.as<float>(), ld, st)) return rc;
      } else if (k) {
        // update_centroids over the sharded rows (ivfflat.rs:47-71): global member counts by all-gather (integers),
        // running sums CHAINED through the ranks in ascending-range order, division on the last rank, broadcast.
        // A LOCAL failure (a HIP error, a failed launch) is remembered and the rank still walks through every rendezvous
        // of the pass -- its peers are waiting in them -- and all ranks leave together at the agreement behind the broadcast.
        int32_t rc_u = VERS_OK;
        auto local = [&](int32_t rc) { if (rc && !rc_u) rc_u = rc; };
        auto hip_local = [&](hipError_t e, const char* what) { if (e != hipSuccess) { (void)hipGetLastError(); local(fail(VERS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e))); } };
        hip_local(hipStreamSynchronize(st), "hipStreamSynchronize");
        if (int32_t rc = comm_rc(cm->all_gather(cm->ctx, counts, counts_all.p, (uint64_t)k * 4), "all_gather")) return rc;
        hipLaunchKernelGGL(sum_counts_kernel, dim3((k + 255) / 256), dim3(256), 0, st, counts_all.as<uint32_t>(), W, k, counts_g.as<uint32_t>());
        hip_local(hipGetLastError(), "sum_counts_kernel");
        if (me == 0) hip_local(hipMemsetAsync(S.p, 0, cbytes, st), "hipMemsetAsync");
        else {
          hip_local(hipStreamSynchronize(st), "hipStreamSynchronize");
          if (int32_t rc = comm_rc(cm->recv(cm->ctx, S.p, (uint64_t)k * ld * 4, me - 1), "recv")) return rc;
        }
        {
          KmTimer t(st, &BuildStats::update_ms);  // (this rank's share of the chained sums; the hops are the host's collectives)
          local(km_update_sums(X, ldx, h->d, sorted.as<uint32_t>(), starts, k, S.as<float>(), ld, st));
          if (me + 1 == W) local(km_finish_centroids(S.as<float>(), counts_g.as<uint32_t>(), k, ld, Cn.as<float>(), st));
        }
        hip_local(hipStreamSynchronize(st), "hipStreamSynchronize");
        if (me + 1 < W)
          if (int32_t rc = comm_rc(cm->send(cm->ctx, S.p, (uint64_t)k * ld * 4, me + 1), "send")) return rc;
        if (int32_t rc = comm_rc(cm->broadcast(cm->ctx, Cn.p, (uint64_t)k * ld * 4, W - 1), "broadcast")) return rc;
        if (int32_t rc = agree(rc_u, "update_centroids")) return rc;
      }
      if (int32_t rc = km_differs(C.as<float>(), Cn.as<float>(), (uint64_t)k * ld, flag_dev, st)) return rc;
      uint32_t differs = 0;
      VERS_HIP_TRY(hipMemcpyAsync(&differs, flag_dev, 4, hipMemcpyDeviceToHost, st));
      VERS_HIP_TRY(hipStreamSynchronize(st));
      ++iters;
      if (!differs) break;  // ivfflat.rs:91-93: bitwise equal -> keep the OLD centroids and stop
      std::swap(C.p, Cn.p);
      std::swap(C.cap, Cn.cap);
    }
    if (out_iterations) out_iterations[a] = iters;
    if (int32_t rc = agree(assign_pass(C.as<float>(), assign.as<uint32_t>(), mind.as<float>()), "the final assign_to_clusters")) return rc;
    // calculate_kmeans_cost (ivfflat.rs:138-149): one left-to-right f32 fold over ALL points -- chained like the sums
    const float* fold_init = nullptr;
    if (multi && me > 0) {
      VERS_HIP_TRY(hipStreamSynchronize(st));
      if (int32_t rc = comm_rc(cm->recv(cm->ctx, cost_in, 4, me - 1), "recv")) return rc;
      fold_init = cost_in;
    }
    int32_t rc_fold;
    {
      KmTimer t(st, &BuildStats::cost_ms);
      rc_fold = km_cost_fold(mind.as<float>(), n, fold_init, cost_dev, st);
    }
    if (!multi && rc_fold) return rc_fold;  // (sharded: the rank still hands a word on and is heard at the agreement below)
    uint32_t stw = 0;
    if (multi) {
      VERS_HIP_TRY(hipStreamSynchronize(st));
      if (me + 1 < W)
        if (int32_t rc = comm_rc(cm->send(cm->ctx, cost_dev, 4, me + 1), "send")) return rc;
      if (int32_t rc = comm_rc(cm->broadcast(cm->ctx, cost_dev, 4, W - 1), "broadcast")) return rc;
      // a NaN distance anywhere fails the build everywhere (the reference panics)
      VERS_HIP_TRY(hipMemcpyAsync(ctl.p, h->km.status.p, 16, hipMemcpyDeviceToDevice, st));
      VERS_HIP_TRY(hipStreamSynchronize(st));
      if (int32_t rc = comm_rc(cm->all_gather(cm->ctx, ctl.p, ctl_all.p, 16), "all_gather")) return rc;
      std::vector<uint32_t> sw(4 * (size_t)W);
      VERS_HIP_TRY(hipMemcpy(sw.data(), ctl_all.p, 16 * (size_t)W, hipMemcpyDeviceToHost));
      for (uint32_t r = 0; r < W; ++r) stw |= sw[4 * r];
      if (int32_t rc = agree(rc_fold, "calculate_kmeans_cost")) return rc;
    }
    float cost = 0.0f;
    VERS_HIP_TRY(hipMemcpyAsync(&cost, cost_dev, 4, hipMemcpyDeviceToHost, st));
    if (!multi) VERS_HIP_TRY(hipMemcpyAsync(&stw, h->km.status.p, 4, hipMemcpyDeviceToHost, st));
    VERS_HIP_TRY(hipStreamSynchronize(st));
    if ((stw & 1u) && k >= 2) {
      VERS_HIP_TRY(hipMemset(h->km.status.p, 0, 16));
      return fail(VERS_ERR_NAN, "NaN distance in assign_to_clusters (reference panics)");
    }
    if (cost < best) {  // strict: the first best attempt wins (ivfflat.rs:116)
      best = cost;
      *out_kept = 1;
      VERS_HIP_TRY(hipMemcpyAsync(bestC.p, C.p, cbytes, hipMemcpyDeviceToDevice, st));
      VERS_HIP_TRY(hipMemcpyAsync(best_assign.p, assign.p, (n ? n : 1) * 4, hipMemcpyDeviceToDevice, st));
    }
  }
  km_timers_collect();
  *out_cost = best;
  if (*out_kept) {
    if (int32_t rc = h->centroids.reserve(cbytes)) return rc;
    VERS_HIP_TRY(hipMemcpyAsync(h->centroids.p, bestC.p, cbytes, hipMemcpyDeviceToDevice, st));
  }
  VERS_HIP_TRY(hipStreamSynchronize(st));
  return VERS_OK;
}

int32_t build_common(vers_ivf* h, const float* X, uint32_t ldx, uint64_t n, const BuildShard& sh_in, uint64_t num_clusters, uint64_t num_attempts,
                     uint64_t max_iterations, const uint64_t* init_indices, float* out_centroids, uint64_t c_stride_bytes,
                     uint64_t* out_assignments, float* out_cost, int32_t* out_kept, uint64_t* out_iterations) {
  const uint32_t k = (uint32_t)num_clusters;
  DevBuf best_assign;
  float cost = INFINITY;
  int32_t kept = 0;
  // the agreement's words exist before the first collective of the build (see Agreement)
  Agreement ag;
  if (int32_t rc = ag.init(sh_in.comm, sh_in.world)) return rc;
  BuildShard sh = sh_in;
  sh.agree = &ag;
  if (int32_t rc = run_build(h, X, ldx, n, sh, k, num_attempts, max_iterations, init_indices, best_assign, &cost, &kept,
                             out_iterations, nullptr))
    return rc;
  if (out_cost) *out_cost = cost;
  if (out_kept) *out_kept = kept;
  if (!kept) {
    // nothing kept: centroids and assignments stay EMPTY, ids = num_clusters empty lists (ivfflat.rs:109-110,123)
    h->k = 0;
    h->n_total = 0;
    h->cap_rows = 0;
    h->max_len = 0;
    h->h_len.clear(); h->h_off.clear(); h->h_cap.clear();
    return VERS_OK;
  }
  if (sh.comm != nullptr && sh.world > 1) {
    if (int32_t rc = install_index_sharded(h, X, ldx, n, sh, best_assign.as<uint32_t>(), k, nullptr)) return rc;
  } else {
    if (int32_t rc = install_index(h, X, ldx, n, best_assign.as<uint32_t>(), k, nullptr)) return rc;
  }
  if (out_centroids && k)
    VERS_HIP_TRY(hipMemcpy2D(out_centroids, (size_t)c_stride_bytes, h->centroids.p, (size_t)h->ldx * 4, (size_t)h->d * 4, k,
                             hipMemcpyDeviceToHost));
  if (out_assignments && n) {
    DevBuf a64;
    if (int32_t rc = a64.reserve(n * 8)) return rc;
    hipLaunchKernelGGL(u32_to_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, best_assign.as<uint32_t>(), n,
                       a64.as<uint64_t>());
    VERS_HIP_TRY(hipGetLastError());
    VERS_HIP_TRY(hipMemcpy(out_assignments, a64.p, n * 8, hipMemcpyDeviceToHost));
  }
  return VERS_OK;
}

// ---- search ---------------------------------------------------------------------------------------
template <int QG, bool SEQ_IDS>
int32_t launch_seg_scan(vers_ivf* h, const SegSrc<QG, SEQ_IDS>& src, uint32_t n_items, int metric, hipStream_t st,
                        const uint64_t* lower = nullptr) {
  ScanParams p;
  p.ld = h->ld;
  p.n_chunks = h->ld / kChunk;
  p.k = src.k;
  p.status = W->st_word();
  p.debug = 0;
  p.stamps = nullptr;
  p.next_quad = nullptr;
  p.bounds = nullptr;  // items of a query are concurrent: nothing to prune, and the atomics would contend
  p.lower = lower;     // (P > 64 ranked lists: 64 ranks per pass)
  const size_t lds = scan_lds_bytes(QG, h->ld);
  uint32_t blocks = (n_items + kWavesPerBlock - 1) / kWavesPerBlock;
  const uint32_t max_blocks = (uint32_t)h->n_cu * scan_blocks_per_cu(QG, h->ld);
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks == 0) blocks = 1;
  if (metric == 0) {
    if (int32_t rc = scan_prepare_launch(scan_kernel<QG, 0, SegSrc<QG, SEQ_IDS>>, lds)) return rc;
    hipLaunchKernelGGL((scan_kernel<QG, 0, SegSrc<QG, SEQ_IDS>>), dim3(blocks), dim3(kWave * kWavesPerBlock), lds, st, src, p);
  } else {
    if (int32_t rc = scan_prepare_launch(scan_kernel<QG, 1, SegSrc<QG, SEQ_IDS>>, lds)) return rc;
    hipLaunchKernelGGL((scan_kernel<QG, 1, SegSrc<QG, SEQ_IDS>>), dim3(blocks), dim3(kWave * kWavesPerBlock), lds, st, src, p);
  }
  VERS_HIP_TRY(hipGetLastError());
  return VERS_OK;
}

// queries (device, pitch ldq_in) -> W->qp [b][ldq] zero padded; returns the pointer/pitch to use
int32_t stage_plain_queries(vers_ivf* h, const float* q_dev, uint64_t ldq_in, uint32_t b, const float** q_out,
                            hipStream_t st) {
  // rows padded to the GEMM tile (the MFMA pre-selection reads whole 128-row tiles; extra rows are ignored)
  if (int32_t rc = W->qp.reserve((size_t)round_up(b, kGemmBM) * h->ldq * sizeof(float))) return rc;
  if (int32_t rc = launch_stage_queries(q_dev, ldq_in, h->d, W->qp.as<float>(), h->ldq, b, 1, st)) return rc;
  *q_out = W->qp.as<float>();
  return VERS_OK;
}

// finished-blocks counter of coarse1_kernel: zero once, the kernel leaves it zero
int32_t ensure_block_counters(vers_ivf* h, hipStream_t st) {
  if (W->c1_ctr.p) return VERS_OK;
  if (int32_t rc = W->c1_ctr.reserve(16)) return rc;
  VERS_HIP_TRY(hipMemsetAsync(W->c1_ctr.p, 0, 16, st));
  return VERS_OK;
}

// Tuning / A-B knobs of the search path (environment, read ONCE per process; DESIGN.md section 5 "Switches").
struct SearchKnobs {
  int qg = 0;            // VERS_QG: 8 or 16 forces the ordered-chain group width
  int pre_slack = 0;     // VERS_PRE_SLACK: slack keys of the matrix-core lists
  long seg_rows = 0;     // VERS_SEG_ROWS
  int pre_blocks_per_cu = 0;  // VERS_PRE_BLOCKS_PER_CU
  int pre_mode = 1;      // VERS_PRESCAN: 0 ordered chains for batches too, 2 every certificate fails
  bool seg_balanced = true;   // VERS_SEG_BALANCED
  uint32_t hot_ranks = 1;     // VERS_HOT_FIRST
};
inline const SearchKnobs& knobs() {
  static const SearchKnobs k = [] {
    SearchKnobs s;
    auto geti = [](const char* n, long dflt) { const char* e = getenv(n); return e ? atol(e) : dflt; };
    s.qg = (int)geti("VERS_QG", 0);
    s.pre_slack = (int)geti("VERS_PRE_SLACK", 0);
    s.seg_rows = geti("VERS_SEG_ROWS", 0);
    s.pre_blocks_per_cu = (int)geti("VERS_PRE_BLOCKS_PER_CU", 0);
    s.pre_mode = (int)geti("VERS_PRESCAN", 1);
    s.seg_balanced = geti("VERS_SEG_BALANCED", 1) != 0;
    s.hot_ranks = (uint32_t)geti("VERS_HOT_FIRST", 1);
    return s;
  }();
  return k;
}

inline int coarse_mode() {  // VERS_COARSE: 1 = always exact, 2 = every certificate fails
  static const int m = [] { const char* e = getenv("VERS_COARSE"); return e ? atoi(e) : 0; }();
  return m;
}
inline bool coarse_on_matrix_cores(const vers_ivf* h, uint32_t b) { return b >= 32 && coarse_mode() != 1 && !W->ref_deep; }
// batches: MFMA pre-selection + exact re-score + certificate (gemm.hip.h); same output as the exact scan, bit for bit.
// qp: staged queries [round_up(b, kGemmBM)][ldq]; probe_out [b][P].  W->gbuf is the only scratch.
// plan != nullptr: the selection kernel also makes every query's plan (plan.hip.h step 1) in its tail.
int32_t coarse_mfma(vers_ivf* h, const float* qp, uint32_t b, uint32_t P, uint64_t* probe_out, hipStream_t st, const PlanQ* plan = nullptr) {
  const uint32_t M_pad = round_up(b, kGemmBM);
  const uint32_t PS = std::min<uint32_t>(kMaxTopK, P + 16);
  if (int32_t rc = W->gbuf.reserve((size_t)M_pad * h->k_pad * sizeof(float))) return rc;
  const bool timed = st != W->ahead_stream || W->ahead_stream == nullptr;  // (the look-ahead stream is not the measured one)
  if (timed) VERS_HIP_TRY(hipEventRecord(W->evc[0], st));
  const __bf16* cs = h->centroids_gs.as<__bf16>();
  VERS_HIP_TRY(launch_gemm<false>((gemm_x3_mask() & 2) != 0, M_pad / kGemmBM, h->k_pad / kGemmBN, st, qp, h->centroids_g.as<float>(),
                                  h->cnorm.as<float>(), h->ldq, h->k_pad, W->gbuf.as<float>(), h->metric, 0, nullptr, nullptr, nullptr, cs,
                                  cs ? cs + (size_t)h->k_pad * h->ldq : nullptr));
  if (timed) VERS_HIP_TRY(hipEventRecord(W->evc[1], st));
  hipLaunchKernelGGL(coarse_select_rescore_kernel, dim3(b), dim3(kWave), 0, st, W->gbuf.as<float>(), h->k_pad, h->k,
                     h->centroids_g.as<float>(), h->ldq, qp, h->ldq, h->ldq, coarse_mode() == 2 ? __builtin_inff() : h->cmax2, P, PS,
                     probe_out, W->st_word(), h->coarse_stat.as<uint32_t>(), h->metric,
                     (scan_debug_flags() & 16u) && W->stamps.p ? W->stamps.as<unsigned long long>() : (unsigned long long*)nullptr,
                     plan ? *plan : PlanQ{});
  VERS_HIP_TRY(hipGetLastError());
  if (timed) { VERS_HIP_TRY(hipEventRecord(W->evc[2], st)); W->evc_valid = true; }
  h->mfma_batches += 1;
  return VERS_OK;
}

// coarse quantiser (ivfflat.rs:155-161): top-P centroids per query as ascending (dist, index) keys in W->probe
// out_n_segs != nullptr: stop after the scan (partial slots in W->cpart) and report the slot count per query --
// the single-query path merges them inside plan1_kernel.
// plan / planned (nullable): when the ranking runs on the matrix cores the queries' plans are made in the same launch and
// *planned is set; otherwise the caller plans from W->probe (plan_queries_kernel).
int32_t coarse(vers_ivf* h, const float* qp, uint32_t b, uint32_t P, hipStream_t st, uint32_t* out_n_segs = nullptr,
               const PlanQ* plan = nullptr, bool* planned = nullptr) {
  // (the contraction reads whole 128-row tiles: the staged block is padded to them, a caller's block used in place is a whole
  // number of them; the selection keeps P + 16 keys: one per lane)
  if (coarse_on_matrix_cores(h, b) && (qp == W->qp.as<float>() || b % kGemmBM == 0) && P + 16 <= (uint32_t)kMaxTopK) {
    if (int32_t rc = W->probe.reserve((size_t)b * P * sizeof(uint64_t))) return rc;
    if (planned) *planned = plan != nullptr;
    return coarse_mfma(h, qp, b, P, W->probe.as<uint64_t>(), st, plan);
  }
  const int QG = b == 1 ? 1 : 8;
  const uint32_t n_qg = (b + QG - 1) / QG;
  const float* q = qp;
  if (QG != 1) {
    if (int32_t rc = W->qil.reserve((size_t)n_qg * h->ldq * QG * sizeof(float))) return rc;
    if (int32_t rc = launch_stage_queries(qp, h->ldq, h->ldq, W->qil.as<float>(), h->ldq, b, QG, st)) return rc;
    q = W->qil.as<float>();
  }
  const uint32_t target_items = (uint32_t)h->n_cu * scan_blocks_per_cu(QG, h->ld) * kWavesPerBlock;
  uint64_t per = ((uint64_t)h->k * n_qg + target_items - 1) / target_items;
  const uint32_t seg_rows = (uint32_t)std::min<uint64_t>(round_up64(per ? per : 1, kWave), max_seg_rows(h->ld));
  const uint32_t n_segs = (h->k + seg_rows - 1) / seg_rows;
  const uint32_t kw = std::min<uint32_t>(P, kMaxTopK);  // keys per partial slot: one per lane; P > 64 takes ceil(P / 64) passes
  if (int32_t rc = W->cpart.reserve((size_t)b * n_segs * kw * sizeof(uint64_t))) return rc;
  if (int32_t rc = W->probe.reserve((size_t)b * P * sizeof(uint64_t))) return rc;
  if (P > (uint32_t)kMaxTopK)
    if (int32_t rc = W->clower.reserve((size_t)b * sizeof(uint64_t))) return rc;
  const uint32_t n_segs_pad = QG == 1 ? n_segs : round_up(n_segs, 4);
  for (uint32_t rank0 = 0; rank0 < P; rank0 += kMaxTopK) {
    const uint32_t k_pass = std::min<uint32_t>(kMaxTopK, P - rank0);
    const uint64_t* lower = rank0 ? W->clower.as<uint64_t>() : nullptr;
    auto fill = [&](auto& src) {
      src.rows = h->centroids_b.as<float>(); src.n = h->k; src.ld = h->ld; src.seg_rows = seg_rows; src.n_segs = n_segs;
      src.n_segs_pad = n_segs_pad;
      src.queries = q; src.ldq = h->ldq; src.b = b; src.partials = W->cpart.as<uint64_t>(); src.k = k_pass; src.ids = nullptr;
    };
    int32_t rc;
    if (QG == 1) {
      SegSrc<1, false> src; fill(src);
      rc = launch_seg_scan(h, src, n_segs_pad * n_qg, h->metric, st, lower);
    } else {
      SegSrc<8, false> src; fill(src);
      rc = launch_seg_scan(h, src, n_segs_pad * n_qg, h->metric, st, lower);
    }
    if (rc) return rc;
    if (out_n_segs) {  // (single query, P <= 64: plan1_kernel merges the slots itself)
      *out_n_segs = n_segs;
      return VERS_OK;
    }
    hipLaunchKernelGGL(coarse_merge_kernel, dim3(b), dim3(kWave * kMergeWaves), 0, st, W->cpart.as<uint64_t>(), n_segs, k_pass, P, rank0,
                       W->probe.as<uint64_t>(), P > (uint32_t)kMaxTopK ? W->clower.as<uint64_t>() : (uint64_t*)nullptr);
    VERS_HIP_TRY(hipGetLastError());
  }
  return VERS_OK;
}

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
  // scans); VERS_SCAN_DEBUG bit 3 switches them off for A/B runs
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
int32_t launch_scan1(vers_ivf* h, const Scan1Args& a, uint32_t items_bound, hipStream_t st, const uint64_t* lower) {
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
  if (!no_ev) VERS_HIP_TRY(hipEventRecord(W->ev0[slot], st));
  if (h->metric) hipLaunchKernelGGL(scan1_kernel<1>, dim3(blocks), dim3(kWave * kWavesPerBlock), 0, st, a, p);
  else hipLaunchKernelGGL(scan1_kernel<0>, dim3(blocks), dim3(kWave * kWavesPerBlock), 0, st, a, p);
  VERS_HIP_TRY(hipGetLastError());
  if (!no_ev) {
    VERS_HIP_TRY(hipEventRecord(W->ev1[slot], st));
    W->ev_count += 1;
  }
  return VERS_OK;
}

// the matrix-core list scan (prescan.hip.h); timed through the same event ring as launch_ivf_scan
int32_t launch_prescan(vers_ivf* h, const IvfSrc<kPreQ>& src, uint32_t items_bound, uint32_t kp, uint32_t* qflags, uint32_t* quad_ctr,
                       bool shadow, hipStream_t st) {
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
  const size_t lds = prescan_lds_bytes_g(h->ld, kp);
  if (int32_t rc = shadow ? scan_prepare_launch(prescan_kernel_g<true, IvfSrc<kPreQ>>, lds) : scan_prepare_launch(prescan_kernel_g<false, IvfSrc<kPreQ>>, lds)) return rc;
  uint32_t blocks = (items_bound + kWavesPerBlock - 1) / kWavesPerBlock;
  uint32_t per_cu = std::max<uint32_t>(1, std::min<uint32_t>(2, (uint32_t)((160u * 1024u) / lds)));  // 1 at d = 768 (measured: as fast as 2)
  if (knobs().pre_blocks_per_cu > 0) per_cu = (uint32_t)knobs().pre_blocks_per_cu;  // tuning knob
  const uint32_t max_blocks = (uint32_t)h->n_cu * per_cu;
  if (blocks > max_blocks) blocks = max_blocks;
  if (blocks == 0) blocks = 1;
  const uint32_t slot = (uint32_t)(W->ev_count % SearchWs::kEvRing);
  if (W->ev_on) VERS_HIP_TRY(hipEventRecord(W->ev0[slot], st));
  if (shadow) hipLaunchKernelGGL((prescan_kernel_g<true, IvfSrc<kPreQ>>), dim3(blocks), dim3(kWave * kPreWavesG), lds, st, src, p);
  else hipLaunchKernelGGL((prescan_kernel_g<false, IvfSrc<kPreQ>>), dim3(blocks), dim3(kWave * kPreWavesG), lds, st, src, p);
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
  const int ref_mode = nprobe == 0;
  { const int ev = scan_events_ref().load(); W->ev_on = ev == 1 || (ev == 2 && b > 1); }
  // reference mode ranks the 48 nearest lists (48 + 16 slack = one key per lane in the MFMA pre-selection);
  // a spill deeper than that is refused (kStSpillTooDeep) -- it needs > 47 consecutive near-empty lists
  // (host-pointer calls retry deeper: 16, 48, 64 and finally ALL lists, W->ref_all -- the reference walks as far as it must)
  uint32_t P_ref = W->ref_all ? h->k : (W->ref_deep ? 64u : (W->ref_shallow ? 16u : 48u));
  // a device-pointer call cannot retry: it ranks as many lists as the list lengths can make the walk need (vers_ivf::len_asc_prefix)
  // -- 48 unless the index has that many near-empty lists; then the exact ranking runs 64 ranks per pass
  if (ref_mode && W->st_slot == 0) P_ref = std::max<uint32_t>(P_ref, h->lists_that_always_suffice(top_k));
  const uint32_t P = ref_mode ? std::min<uint32_t>(h->k, P_ref) : std::min<uint32_t>(nprobe, h->k);
  // one key per lane is the width of every list in the kernels: more ranked lists (P > 64) or more results (top_k > 64)
  // are produced 64 ranks per pass (ScanParams::lower), on the ordered-chain kernels
  const bool one1 = b == 1 && P <= (uint32_t)kMaxTopK;  // single query: coarse merge + plan fused in plan1_kernel
  // ... and the coarse scan with them in coarse1_kernel (VERS_COARSE1=0: the ordered-chain scan + plan1_kernel, for A/B runs)
  static const bool c1_on = [] { const char* e = getenv("VERS_COARSE1"); return e ? atoi(e) != 0 : true; }();
  const bool one1_fused = one1 && c1_on;
  if (one1_fused) {
    if (int32_t rc = W->cpart.reserve((size_t)((h->k + kWave - 1) / kWave) * P * sizeof(uint64_t))) return rc;
    if (int32_t rc = W->probe.reserve((size_t)P * sizeof(uint64_t))) return rc;
    if (int32_t rc = ensure_block_counters(h, st)) return rc;
  }
  const float* qp = nullptr;
  const uint64_t* probe = nullptr;
  SearchWs::CoarseAhead* took = nullptr;
  for (auto& a : W->ahead)
    if (a.valid && !ref_mode && a.q_dev == q_dev && a.ldq_in == ldq_in && a.b == b && a.P == P) took = &a;
  {
    std::lock_guard<std::mutex> lk(h->pool_mu);
    if (h->pending.set && h->pending.q_dev == q_dev && h->pending.ldq_in == ldq_in && h->pending.b == b) h->pending.set = false;  // this very batch: computed inline below
  }
  // geometry of the list scan
  const uint64_t n_pj = (uint64_t)b * P;
  const uint64_t pairs_est = ref_mode ? b : n_pj;
  const uint64_t lists_est = std::min<uint64_t>(h->k, pairs_est);
  // query-group width of the list scan: a list shared by more queries than one group holds is streamed
  // once per group, so pick the width from the expected queries per list
  int QG = (b == 1 || pairs_est < 2 * lists_est) ? 1 : (pairs_est >= 6 * lists_est ? 16 : 8);
  if (QG != 1 && (knobs().qg == 8 || knobs().qg == 16)) QG = knobs().qg;  // tuning knob
  // Batches in nprobe mode: the list scan runs on the matrix cores with an exact finish (prescan.hip.h); same bits.
  // VERS_PRESCAN=0 keeps the ordered-chain scan, =2 makes every certificate fail (exercises the exact fallback).
  const int pre_mode = knobs().pre_mode;
  // slack of 10 keys: at cfg3 a slack of 6 left ~2 of 1024 queries uncertified per batch, 10 none
  uint32_t kp = std::min<uint32_t>(kPreMaxKp, std::max<uint32_t>(top_k + 10, top_k + top_k / 2));
  // fp16 shadow rows: the certificate window is ~2x the f32 rows' (measured residual, prescan.hip.h)
  bool use_shadow = shadow_mode() != 0 && !h->shadow_off && h->shadow_valid && h->rows_bf.p != nullptr &&
                    h->rows_bf.cap >= h->cap_rows * (size_t)h->ld * sizeof(uint16_t);
  if (use_shadow && h->fail_watch && h->shadow_queries >= 256) {  // (lags by the batches still in flight: errs on the side of keeping it)
    const uint32_t failed = *reinterpret_cast<volatile uint32_t*>(h->fail_watch);
    if ((uint64_t)failed * 8 > h->shadow_queries) { h->shadow_off = true; use_shadow = false; }
  }
  // (fp16 rows: the window is ~2x the f32 one.  At cfg3 a slack of 10 left ~0.5 of 1024 queries per batch uncertified, 16 none)
  if (use_shadow) kp = std::min<uint32_t>(kPreMaxKp, top_k + std::max<uint32_t>(24, top_k));
  if (knobs().pre_slack > 0) kp = std::min<uint32_t>(kPreMaxKp, top_k + (uint32_t)knobs().pre_slack);  // tuning knob
  const bool use_pre = QG != 1 && !ref_mode && pre_mode != 0 && top_k + 6 <= kPreMaxKp && P <= (uint32_t)kMaxTopK &&
                       prescan_lds_bytes_g(h->ld, kp) <= 160u * 1024u;  // the query block of 32 padded queries must fit LDS
  if (use_pre) QG = kPreQ;
  const uint32_t k_keep = use_pre ? kp : std::min<uint32_t>(top_k, kMaxTopK);
  // 64 result ranks per pass; no pass beyond the rows the index holds (top_k = 100000 on 1000 rows: 16 passes, not 1563)
  const uint32_t n_pass = use_pre ? 1u : (uint32_t)((std::min<uint64_t>(top_k, std::max<uint64_t>(1, h->n_total)) + kMaxTopK - 1) / kMaxTopK);
  const uint64_t groups_bound = QG == 1 ? n_pj : (n_pj / QG + std::min<uint64_t>(h->k, n_pj));
  uint32_t seg_rows;
  const uint64_t avg_len_all = std::max<uint64_t>(1, h->n_total / std::max<uint32_t>(1, h->k));
  if (b == 1) seg_rows = kWave;  // (cut finer below when one query probes very many lists)
  else {
    const uint64_t groups_est = std::max<uint64_t>(1, std::max<uint64_t>(pairs_est / QG, lists_est));
    // ~160 items per CU: short segments give every (list, query group) several quads and even out the tail
    // (measured at N=10M/nlist=4096/batch=1024: 256-row segments 9.0 ms, 640-row 10.6 ms)
    const uint64_t segs_wanted = std::max<uint64_t>(1, ((uint64_t)h->n_cu * 160 + groups_est - 1) / groups_est);
    const uint64_t avg_len = std::max<uint64_t>(1, h->n_total / std::max<uint32_t>(1, h->k));
    // whole tile PAIRS (the batched kernel walks two tiles per step)
    seg_rows = (uint32_t)round_up64(std::max<uint64_t>(1, (avg_len + segs_wanted - 1) / segs_wanted), QG == 1 ? kWave : 2 * kWave);
  }
  // matrix-core scan: an average list is one quad of items (per-item set-up and the block's barriers amortise over
  // ~10 tiles; measured at cfg3: 640-row targets beat 256- and 1024-row ones)
  if (use_pre) seg_rows = (uint32_t)round_up64(std::max<uint64_t>(256, (avg_len_all + 3) / 4), kWave);
  if (use_pre) {
    // ... unless that leaves fewer than ~8 quads per CU -- the lists sharded over GPUs: 2.7 at 8 ranks, the last third of the
    // launch half empty.  Then the lists are cut finer; the scan hands quads out in RUNS and stages once per run of one list
    // (prescan_kernel_g), so the finer cut costs no staging while work is plentiful and balances the tail.
    const uint64_t lists_here = std::max<uint64_t>(1, lists_est / std::max<uint32_t>(1, h->world));
    const uint64_t quads_est = lists_here * std::max<uint64_t>(1, pairs_est / std::max<uint64_t>(1, lists_est * kPreQ));
    // (measured, same box, 8 ranks: 392 us with whole-list quads, 425 with four quads per list, 0.497 / 0.523 / 0.537 ms per step
    // with three batches in flight at 1 / 2 / 4 -- a short item pays its pipeline fill and its waves' waits for each other
    // whatever the staging costs: OFF by default, VERS_FINE_QUADS=2|4 to try)
    static const int fine_max = [] { const char* e = getenv("VERS_FINE_QUADS"); return e ? atoi(e) : 1; }();
    uint32_t fine = 1;
    while ((int)fine < fine_max && quads_est * fine < 8ull * (uint64_t)h->n_cu && seg_rows / (2 * fine) >= 128) fine *= 2;
    seg_rows = (uint32_t)round_up64(seg_rows / fine, kWave);
  }
  if (knobs().seg_rows > 0) seg_rows = (uint32_t)round_up64(std::max(64l, knobs().seg_rows), kWave);  // tuning knob
  // matrix-core scan: per-list balanced segments of about seg_rows rows (list_seg_rows)
  // per-list balanced segments (list_seg_rows): same-box A/B at cfg3 5.96 ms vs 6.27 ms with fixed 640-row segments;
  // VERS_SEG_BALANCED=0 switches them off
  const bool seg_balanced = knobs().seg_balanced;
  const uint32_t seg_target = use_pre && seg_balanced ? seg_rows : 0u;
  // segments per list at most; partial slots per (query, probe): one per segment, or one per QUAD of segments (matrix-core scan)
  const uint32_t S_seg = seg_target ? 4 * std::max<uint32_t>(1, (h->max_len + 4 * seg_target - 1) / (4 * seg_target))
                                    : std::max<uint32_t>(1, (h->max_len + seg_rows - 1) / seg_rows);
  const uint32_t S_max = use_pre ? (S_seg + 3) / 4 : S_seg;
  const uint64_t items_bound = groups_bound * (QG == 1 ? S_seg : round_up(S_seg, 4));
  if (items_bound > 0x7FFFFFFFull) return fail(VERS_ERR_INVALID, "search batch too large");

  const uint32_t k_l = h->k;
  // pj: list, pref, take per (q, j); np per q.   lists: cnt, fill | pair_off, group_off, item_off | totals
  if (int32_t rc = W->pj.reserve((4 * n_pj + b) * sizeof(uint32_t))) return rc;
  // one zero-initialised zone per batch (ONE memset): cnt, fill, hot per list | quad hand-out counter | queue of
  // uncertified queries + its count | non-finite flags per (query, probe)
  const size_t zero_words = 3 * (size_t)k_l + 4 + (use_pre ? (size_t)b + 4 + n_pj : 0);
  if (int32_t rc = W->lists.reserve((zero_words + 3 * (size_t)k_l) * sizeof(uint32_t) + sizeof(GroupTotals) + 64)) return rc;
  if (int32_t rc = W->pairs.reserve(n_pj * sizeof(uint32_t))) return rc;
  if (int32_t rc = W->items.reserve(one1 ? (items_bound + 4) * sizeof(Item1Rec) : std::max<uint64_t>(1, items_bound) * sizeof(ItemDesc))) return rc;
  if (int32_t rc = W->groups.reserve(std::max<uint64_t>(1, groups_bound) * sizeof(GroupDesc))) return rc;
  if (QG != 1 && !use_pre)
    if (int32_t rc = W->qblocks.reserve(groups_bound * h->ldq * QG * sizeof(float))) return rc;
  W->ivf_bounds_off = ((size_t)n_pj * S_max * k_keep + 1) & ~(size_t)1;  // (even: the bounds start 16-byte aligned)
  const size_t part_bytes = (W->ivf_bounds_off + n_pj) * sizeof(uint64_t);  // slots + one bound per (query, probe)
  if (int32_t rc = W->partials.reserve(part_bytes + 16)) return rc;
  uint32_t* pj_list = W->pj.as<uint32_t>();
  uint32_t* pj_pref = pj_list + n_pj;
  uint32_t* pj_take = pj_pref + n_pj;
  uint32_t* np = pj_take + n_pj;
  uint32_t* pj_nq = np + b;
  uint32_t* cnt = W->lists.as<uint32_t>();
  uint32_t* fill = cnt + k_l;
  uint32_t* hot = fill + k_l;  // lists that are the nearest list of some query: scanned first
  uint32_t* quad_ctr = hot + k_l;
  uint32_t* fail_list = quad_ctr + 4;            // (matrix-core scan only)
  uint32_t* qflags = fail_list + b + 4;
  uint32_t* pair_off = cnt + zero_words;
  uint32_t* group_off = pair_off + k_l;
  uint32_t* item_off = group_off + k_l;
  GroupTotals* tot = (GroupTotals*)(((uintptr_t)(item_off + k_l) + 15) & ~(uintptr_t)15);
  W->tot_dev = tot;

  // The per-batch tables are zeroed FIRST: the queries' plans (counts per list, hot marks) are made in the tail of the coarse
  // quantiser's selection kernel when it runs on the matrix cores, by plan_queries_kernel otherwise.
  PlanQ pq;
  pq.b = b; pq.P = P; pq.k_lists = k_l; pq.top_k = top_k; pq.ref_mode = ref_mode;
  pq.list_len = h->list_len.as<uint32_t>(); pq.owner = h->world > 1 ? h->owner.as<uint8_t>() : (const uint8_t*)nullptr; pq.rank = h->rank;
  pq.list_slot = h->list_slot.as<uint32_t>();
  pq.pj_list = pj_list; pq.pj_pref = pj_pref; pq.pj_take = pj_take; pq.np = np; pq.pj_nq = use_pre ? pj_nq : nullptr;
  pq.cnt = cnt; pq.hot = hot; pq.hot_ranks = knobs().hot_ranks; pq.seg_rows = seg_rows; pq.seg_target = seg_target;
  pq.status = W->st_word();
  bool planned = false;
  if (!one1) VERS_HIP_TRY(hipMemsetAsync(cnt, 0, zero_words * sizeof(uint32_t), st));
  uint32_t n_segs_c = 0;
  if (took) {  // staged queries and ranked lists of this batch were computed ahead (vers_ivf_coarse_ahead_dev)
    VERS_HIP_TRY(hipStreamWaitEvent(st, took->ready, 0));
    qp = took->qp.as<float>();
    probe = took->probe.as<uint64_t>();
    took->valid = false;
    h->ahead_used += 1;
  } else {
    for (auto& a : W->ahead)  // W->gbuf is shared with a look-ahead in flight: let it finish first
      if (a.ready_rec) VERS_HIP_TRY(hipStreamWaitEvent(st, a.ready, 0));
    // the caller's block as it is when its layout already is the staged one: no padding columns to zero (d == ldq), the same
    // pitch, and -- the matrix-core contraction reads whole 128-row tiles -- a whole number of tiles (or a single query)
    if (h->d == h->ldq && (reinterpret_cast<uintptr_t>(q_dev) & 15u) == 0 && (b == 1 || (ldq_in == h->ldq && b % kGemmBM == 0))) qp = q_dev;
    else if (int32_t rc = stage_plain_queries(h, q_dev, ldq_in, b, &qp, st)) return rc;
    if (!one1_fused)
      if (int32_t rc = coarse(h, qp, b, P, st, one1 ? &n_segs_c : nullptr, one1 ? nullptr : &pq, &planned)) return rc;
    probe = W->probe.as<uint64_t>();
  }

  if (n_pass > 1)
    if (int32_t rc = W->lower.reserve(n_pj * sizeof(uint64_t))) return rc;
  if (one1) {
    const bool fill_in_kernel = part_bytes <= (size_t(4) << 20);  // (a block fills a few hundred KB faster than a launch costs)
    if (!fill_in_kernel) VERS_HIP_TRY(hipMemsetAsync(W->partials.p, 0xFF, part_bytes, st));
    Plan1Args pa;
    pa.cpart = W->cpart.as<uint64_t>(); pa.n_segs_c = n_segs_c; pa.P = P; pa.k_lists = k_l; pa.top_k = top_k; pa.ref_mode = ref_mode;
    pa.list_len = h->list_len.as<uint32_t>(); pa.owner = h->world > 1 ? h->owner.as<uint8_t>() : (const uint8_t*)nullptr; pa.rank = h->rank;
    pa.seg_rows = seg_rows; pa.probe = W->probe.as<uint64_t>(); pa.pj_list = pj_list; pa.pj_pref = pj_pref; pa.pj_take = pj_take; pa.np = np;
    pa.cnt = cnt; pa.pair_off = pair_off; pa.group_off = group_off; pa.pairs = W->pairs.as<uint32_t>(); pa.items = W->items.as<ItemDesc>();
    pa.groups = W->groups.as<GroupDesc>(); pa.tot = tot; pa.status = W->st_word(); pa.list_slot = h->list_slot.as<uint32_t>();
    pa.ff_begin = reinterpret_cast<u32x4*>(W->partials.p); pa.ff_vec16 = fill_in_kernel ? (uint32_t)((part_bytes + 15) / 16) : 0u;
    pa.recs = W->items.as<Item1Rec>(); pa.slot_off = h->slot_off.as<uint32_t>(); pa.slot_len = h->slot_len.as<uint32_t>(); pa.S_max = S_max;
    if (one1_fused) {  // coarse quantiser + plan in one launch: a block per 64-centroid tile, the last one to finish plans
      Coarse1Args ca;
      ca.cent = h->centroids_b.as<float>(); ca.k = k_l; ca.ld = h->ld; ca.n_chunks = h->ld / kChunk; ca.q = qp; ca.cpart = W->cpart.as<uint64_t>();
      ca.P = P; ca.ctr = W->c1_ctr.as<uint32_t>(); ca.status = W->st_word();
      ca.stamps = nullptr;
      if (scan_debug_flags() & 16u) {
        if (int32_t rc = W->stamps.reserve(512)) return rc;
        ca.stamps = W->stamps.as<unsigned long long>();
        pa.stamps = ca.stamps;
      }
      pa.n_segs_c = (k_l + kWave - 1) / kWave;
      if (h->metric) {
        if (int32_t rc = scan_prepare_launch(coarse1_kernel<1>, kC1LdsBytes)) return rc;
        hipLaunchKernelGGL(coarse1_kernel<1>, dim3(pa.n_segs_c), dim3(kWave * kC1Phase), kC1LdsBytes, st, ca, pa);
      } else {
        if (int32_t rc = scan_prepare_launch(coarse1_kernel<0>, kC1LdsBytes)) return rc;
        hipLaunchKernelGGL(coarse1_kernel<0>, dim3(pa.n_segs_c), dim3(kWave * kC1Phase), kC1LdsBytes, st, ca, pa);
      }
    } else {
      hipLaunchKernelGGL(plan1_kernel, dim3(1), dim3(kWave * kMergeWaves), 0, st, pa);
    }
    VERS_HIP_TRY(hipGetLastError());
  } else {
  // (also a single query with P > 64)
  if (!planned) {
    hipLaunchKernelGGL(plan_queries_kernel, dim3((b + 3) / 4), dim3(256), 0, st, pq, probe);
    VERS_HIP_TRY(hipGetLastError());
  }
  GroupArgs ga;
  ga.b = b; ga.P = P; ga.k_lists = k_l; ga.QG = (uint32_t)QG; ga.seg_rows = seg_rows; ga.seg_target = seg_target;
  ga.slot_len = h->slot_len.as<uint32_t>(); ga.cnt = cnt; ga.hot = hot; ga.fill = fill; ga.pj_list = pj_list;
  ga.pair_off = pair_off; ga.group_off = group_off; ga.item_off = item_off; ga.tot = tot;
  ga.pairs = W->pairs.as<uint32_t>(); ga.items = W->items.as<ItemDesc>(); ga.groups = W->groups.as<GroupDesc>();
  if (use_pre) {  // only the bounds behind the slots (n_pj words of 8 bytes, 16-byte aligned: ivf_bounds_off is even)
    ga.ff_begin = reinterpret_cast<u32x4*>(W->partials.as<uint64_t>() + W->ivf_bounds_off); ga.ff_vec16 = ((size_t)n_pj * 8 + 15) / 16;
  } else {        // ordered-chain scans: ivf_merge_kernel reads every slot -- a full-width fill
    VERS_HIP_TRY(hipMemsetAsync(W->partials.p, 0xFF, part_bytes, st));
    ga.ff_begin = nullptr; ga.ff_vec16 = 0;
  }
  ga.stamps = nullptr;
  if (scan_debug_flags() & 16u) {
    if (int32_t rc = W->stamps.reserve(512)) return rc;
    ga.stamps = W->stamps.as<unsigned long long>();
  }
  // blocks: enough that a block's share of the pairs and lists is small next to the (redundant) prefix sums
  const uint32_t g_blocks = (uint32_t)std::min<uint64_t>(kGroupMaxBlocks, std::max<uint64_t>(1, (n_pj + 2047) / 2048 + k_l / 256));
  const size_t g_lds = k_l <= kGroupTabMax ? 3 * (size_t)k_l * sizeof(uint32_t) : 0;
  if (int32_t rc = scan_prepare_launch(group_scatter_kernel, g_lds)) return rc;
  hipLaunchKernelGGL(group_scatter_kernel, dim3(g_blocks), dim3(kGroupThreads), g_lds, st, ga);
  VERS_HIP_TRY(hipGetLastError());
  if (QG != 1 && !use_pre) {  // (the matrix-core scan gathers its query block from qp while staging it)
    hipLaunchKernelGGL(gather_qblocks_kernel, dim3((unsigned)groups_bound), dim3(256), 0, st, W->groups.as<GroupDesc>(), tot,
                       W->pairs.as<uint32_t>(), P, qp, h->ldq, (uint32_t)QG, W->qblocks.as<float>());
    VERS_HIP_TRY(hipGetLastError());
  }
  }  // batch planning
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
    // partial lists of the exact re-scan (fail_list [b] + its count, qflags [n_pj]: in the zeroed zone above)
    uint32_t fb_blocks = kFallbackBlocks;  // (a power of two; fewer when P x top_k is large: at most 32 MB of partial lists)
    while (fb_blocks > 16 && fallback_part_keys(fb_blocks, P, top_k) * sizeof(uint64_t) > (size_t(32) << 20)) fb_blocks /= 2;
    if (int32_t rc2 = W->fb_part.reserve(fallback_part_keys(fb_blocks, P, top_k) * sizeof(uint64_t))) return rc2;
    if (!W->fb_ctr.p) {  // group counters of fallback_kernel: zero once, the kernel leaves them zero
      if (int32_t rc2 = W->fb_ctr.reserve((2 * kFallbackBlocks + 1) * sizeof(uint32_t))) return rc2;
      VERS_HIP_TRY(hipMemsetAsync(W->fb_ctr.p, 0, (2 * kFallbackBlocks + 1) * sizeof(uint32_t), st));
    }
    if (int32_t rc2 = launch_prescan(h, src, (uint32_t)items_bound, kp, qflags, quad_ctr, use_shadow, st)) return rc2;
    if (int32_t rc2 = start_pending_ahead(h, st)) return rc2;  // the next batch's coarse quantiser: under this batch's exact finish
    RescoreArgs a;
    a.partials = W->partials.as<uint64_t>(); a.P = P; a.S_max = S_max; a.kp = kp; a.top_k = top_k; a.d_pad = h->ld;
    a.pj_list = pj_list; a.pj_pref = pj_pref; a.pj_nq = pj_nq; a.list_off = h->slot_off.as<uint32_t>(); a.row_ids = h->row_ids.as<uint32_t>();
    a.rows = h->rows.as<float>(); a.rows_rm = h->rows_rm.as<float>(); a.ld = h->ld; a.qp = qp; a.ldq = h->ldq; a.xmax2_bits = h->pre_misc.as<uint32_t>();
    a.qflags = qflags; a.metric = h->metric; a.force_fail = pre_mode == 2; a.shadow = use_shadow ? 1 : 0; a.debug = scan_debug_flags(); a.fail_list = fail_list; a.stats = h->pre_misc.as<uint32_t>() + 1;
    a.status = W->st_word(); a.out_ids = out_ids; a.out_dist = out_dist; a.out_count = out_count; a.out_keys = out_keys;
    const int stage_rows = rescore_lds_bytes(h->ld, true) <= 144u * 1024u ? 1 : 0;
    const size_t rs_lds = rescore_lds_bytes(h->ld, stage_rows != 0);
    if (int32_t rc2 = scan_prepare_launch(ivf_rescore_kernel, rs_lds)) return rc2;
    hipLaunchKernelGGL(ivf_rescore_kernel, dim3(b), dim3(kWave * kRescoreWaves), rs_lds, st, a, stage_rows);
    VERS_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(fallback_kernel, dim3(fb_blocks), dim3(kWave * kMergeWaves), 0, st, a, (const uint32_t*)h->slot_len.as<uint32_t>(),
                       (const uint32_t*)fail_list, (const uint32_t*)(fail_list + b), W->fb_part.as<uint64_t>(), W->fb_ctr.as<uint32_t>(),
                       use_shadow ? h->fail_watch : (uint32_t*)nullptr);
    VERS_HIP_TRY(hipGetLastError());
    W->last_pre.valid = true; W->last_pre.b = b; W->last_pre.P = P; W->last_pre.S_max = S_max; W->last_pre.kp = kp; W->last_pre.top_k = top_k;
    W->last_pre.qp = qp; W->last_pre.shadow = use_shadow ? 1 : 0;
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
      rc = launch_scan1(h, sa, (uint32_t)items_bound, st, lower);
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

// Stage the queries of a coming batch and rank its lists on the side stream (see SearchWs::CoarseAhead).
int32_t coarse_ahead_locked(vers_ivf* h, const float* q_dev, uint64_t ldq_in, uint32_t b, uint32_t nprobe, hipStream_t st) {
  if (b == 0 || nprobe == 0 || h->k == 0 || !coarse_on_matrix_cores(h, b)) return VERS_OK;  // nothing to gain: the search does it itself
  const uint32_t P = std::min<uint32_t>(nprobe, h->k);
  if (P + 16 > (uint32_t)kMaxTopK) return VERS_OK;  // ranked exactly inside the search (more lists than the matrix-core selection holds)
  if (!W->ahead_stream) {
    VERS_HIP_TRY(hipStreamCreateWithFlags(&W->ahead_stream, hipStreamNonBlocking));
    VERS_HIP_TRY(hipEventCreateWithFlags(&W->ahead_in, hipEventDisableTiming));
    for (auto& a : W->ahead) {
      VERS_HIP_TRY(hipEventCreateWithFlags(&a.ready, hipEventDisableTiming));
      VERS_HIP_TRY(hipEventCreateWithFlags(&a.freed, hipEventDisableTiming));
    }
  }
  for (auto& a : W->ahead)
    if (a.valid && a.q_dev == q_dev && a.ldq_in == ldq_in && a.b == b && a.P == P) return VERS_OK;  // already prepared
  SearchWs::CoarseAhead& a = W->ahead[W->ahead_next];
  W->ahead_next ^= 1u;
  hipStream_t side = W->ahead_stream;
  // after everything already queued on the caller's stream (whatever produced the queries; any search still using
  // W->gbuf), and after the search that read this slot last
  VERS_HIP_TRY(hipEventRecord(W->ahead_in, st));
  VERS_HIP_TRY(hipStreamWaitEvent(side, W->ahead_in, 0));
  if (a.freed_rec) VERS_HIP_TRY(hipStreamWaitEvent(side, a.freed, 0));
  a.valid = false;
  if (int32_t rc = a.qp.reserve((size_t)round_up(b, kGemmBM) * h->ldq * sizeof(float))) return rc;
  if (int32_t rc = a.probe.reserve((size_t)b * P * sizeof(uint64_t))) return rc;
  if (int32_t rc = launch_stage_queries(q_dev, ldq_in, h->d, a.qp.as<float>(), h->ldq, b, 1, side)) return rc;
  if (int32_t rc = coarse_mfma(h, a.qp.as<float>(), b, P, a.probe.as<uint64_t>(), side)) return rc;
  VERS_HIP_TRY(hipEventRecord(a.ready, side));
  a.ready_rec = true;
  a.q_dev = q_dev; a.ldq_in = ldq_in; a.b = b; a.P = P; a.valid = true;
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

int32_t relayout(vers_ivf* h) {
  const uint32_t k = h->k;
  std::vector<uint32_t> noff(k), ncap(k);
  uint64_t off = 0;
  for (uint32_t c = 0; c < k; ++c) {
    const uint32_t len = h->h_len[c];
    ncap[c] = h->h_owner[c] == h->rank ? round_up(len + std::max<uint32_t>(64u, len / 8u), 64u) : 0u;
    noff[c] = (uint32_t)off;
    off += ncap[c];
    if (off > 0xFFFFFFFFull) return fail(VERS_ERR_INVALID, "more than 2^32-1 storage rows on one GPU");
  }
  DevBuf nrows, nids;
  if (int32_t rc = nrows.reserve((off ? off : 1) * (size_t)h->ld * sizeof(float))) return rc;
  if (int32_t rc = nids.reserve((off ? off : 1) * sizeof(uint32_t))) return rc;
  VERS_HIP_TRY(hipMemset(nids.p, 0xFF, (off ? off : 1) * sizeof(uint32_t)));
  for (uint32_t c = 0; c < k; ++c) {
    if (!h->h_len[c] || h->h_owner[c] != h->rank) continue;
    // lists start on tile boundaries, so whole 64-row tiles move as they are
    VERS_HIP_TRY(hipMemcpyAsync(nrows.as<float>() + (size_t)noff[c] * h->ld, h->rows.as<float>() + (size_t)h->h_off[c] * h->ld,
                                (size_t)round_up(h->h_len[c], 64) * h->ld * sizeof(float), hipMemcpyDeviceToDevice, nullptr));
    VERS_HIP_TRY(hipMemcpyAsync(nids.as<uint32_t>() + noff[c], h->row_ids.as<uint32_t>() + h->h_off[c],
                                (size_t)h->h_len[c] * sizeof(uint32_t), hipMemcpyDeviceToDevice, nullptr));
  }
  VERS_HIP_TRY(hipDeviceSynchronize());
  std::swap(h->rows.p, nrows.p); std::swap(h->rows.cap, nrows.cap);
  std::swap(h->row_ids.p, nids.p); std::swap(h->row_ids.cap, nids.cap);
  h->h_off = noff; h->h_cap = ncap; h->cap_rows = off;
  VERS_HIP_TRY(hipMemcpy(h->list_off.p, h->h_off.data(), (size_t)k * 4, hipMemcpyHostToDevice));
  {
    std::vector<uint32_t> so(k ? k : 1);
    for (uint32_t c = 0; c < k; ++c) so[h->h_slot[c]] = h->h_off[c];
    if (k) VERS_HIP_TRY(hipMemcpy(h->slot_off.p, so.data(), (size_t)k * 4, hipMemcpyHostToDevice));
  }
  if (int32_t rc = refresh_norms(h, 0, h->cap_rows, nullptr)) return rc;
  VERS_HIP_TRY(hipDeviceSynchronize());
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
    VERS_HIP_TRY(hipStreamSynchronize(W->io_stream));
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

}  // namespace

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
  delete h;
  return VERS_OK;
}

int32_t vers_ivf_build(vers_ivf_t* h, const float* rows, uint64_t n, uint64_t row_stride_bytes, uint64_t num_clusters,
                       uint64_t num_attempts, uint64_t max_iterations, const uint64_t* init_indices, float* out_centroids,
                       uint64_t c_stride_bytes, uint64_t* out_assignments, float* out_cost, int32_t* out_kept,
                       uint64_t* out_iterations) {
  if (!h || (n && !rows) || row_stride_bytes < (uint64_t)(h ? h->d : 0) * 4 || row_stride_bytes % 4 ||
      (num_attempts * num_clusters && !init_indices) || n > 0xFFFFFFFFull || num_clusters > 0xFFFFFFFFull ||
      (out_centroids && num_clusters && c_stride_bytes < (uint64_t)h->d * 4))
    return fail(VERS_ERR_INVALID, "vers_ivf_build: bad arguments");
  std::unique_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  DevBuf X;
  if (int32_t rc = X.reserve((n ? n : 1) * (size_t)h->ldx * sizeof(float))) return rc;
  if (n) {
    if (h->ldx != h->d) VERS_HIP_TRY(hipMemset(X.p, 0, n * (size_t)h->ldx * sizeof(float)));
    VERS_HIP_TRY(hipMemcpy2D(X.p, (size_t)h->ldx * 4, rows, row_stride_bytes, (size_t)h->d * 4, n, hipMemcpyHostToDevice));
  }
  BuildShard one;
  one.n_total = n;
  return build_common(h, X.as<float>(), h->ldx, n, one, num_clusters, num_attempts, max_iterations, init_indices, out_centroids,
                      c_stride_bytes, out_assignments, out_cost, out_kept, out_iterations);
}

int32_t vers_ivf_build_dev(vers_ivf_t* h, const float* rows_dev, uint64_t n, uint64_t ld_floats, uint64_t num_clusters,
                           uint64_t num_attempts, uint64_t max_iterations, const uint64_t* init_indices, float* out_centroids,
                           uint64_t c_stride_bytes, uint64_t* out_assignments, float* out_cost, int32_t* out_kept,
                           uint64_t* out_iterations) {
  if (!h || (n && !rows_dev) || ld_floats < (h ? h->d : 0) || ld_floats % 4 || ld_floats > 0x3FFFFFFFull ||
      (num_attempts * num_clusters && !init_indices) || n > 0xFFFFFFFFull || num_clusters > 0xFFFFFFFFull ||
      (out_centroids && num_clusters && c_stride_bytes < (uint64_t)h->d * 4))
    return fail(VERS_ERR_INVALID, "vers_ivf_build_dev: bad arguments (ld_floats must be >= d and a multiple of 4)");
  std::unique_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  BuildShard one;
  one.n_total = n;
  return build_common(h, rows_dev, (uint32_t)ld_floats, n, one, num_clusters, num_attempts, max_iterations, init_indices, out_centroids, c_stride_bytes,
                      out_assignments, out_cost, out_kept, out_iterations);
}

int32_t vers_ivf_build_sharded_dev(vers_ivf_t* h, const float* rows_dev, uint64_t n_local, uint64_t ld_floats, uint64_t row_begin,
                                   uint64_t n_total, uint64_t num_clusters, uint64_t num_attempts, uint64_t max_iterations,
                                   const uint64_t* init_indices, const vers_comm_t* comm, uint64_t* out_assignments_local, float* out_cost,
                                   int32_t* out_kept, uint64_t* out_iterations) {
  if (!h || !comm || (n_local && !rows_dev) || ld_floats < (h ? h->d : 0) || ld_floats % 4 || ld_floats > 0x3FFFFFFFull ||
      (num_attempts * num_clusters && !init_indices) || n_total > 0xFFFFFFFFull || n_local > n_total || row_begin > n_total - n_local ||
      num_clusters > 0xFFFFFFFFull)
    return fail(VERS_ERR_INVALID, "vers_ivf_build_sharded_dev: bad arguments (ld_floats must be >= d and a multiple of 4; vec ids are 32-bit)");
  if (comm->world == 0 || comm->world > 255 || comm->rank >= comm->world ||
      (comm->world > 1 && (!comm->all_gather || !comm->send || !comm->recv || !comm->broadcast || !comm->all_to_all_v)))
    return fail(VERS_ERR_INVALID, "vers_ivf_build_sharded_dev: incomplete vers_comm_t");
  std::unique_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  BuildShard sh;
  sh.comm = comm->world > 1 ? comm : nullptr;
  sh.rank = comm->rank; sh.world = comm->world; sh.row_begin = row_begin; sh.n_total = n_total;
  if (comm->world == 1 && (row_begin != 0 || n_local != n_total)) return fail(VERS_ERR_INVALID, "vers_ivf_build_sharded_dev: a single rank must hold every row");
  if (comm->world == 1) { h->rank = 0; h->world = 1; }
  return build_common(h, rows_dev, (uint32_t)ld_floats, n_local, sh, num_clusters, num_attempts, max_iterations, init_indices, nullptr, 0,
                      out_assignments_local, out_cost, out_kept, out_iterations);
}

int32_t vers_set_option(const char* name, int64_t value) {
  if (!name) return fail(VERS_ERR_INVALID, "vers_set_option: null name");
  if (std::strcmp(name, "gemm_x3") == 0) { set_gemm_x3_mask((int)value); return VERS_OK; }
  if (std::strcmp(name, "shadow") == 0) { shadow_mode_ref().store(value != 0 ? 1 : 0); return VERS_OK; }
  if (std::strcmp(name, "scan_events") == 0) { scan_events_ref().store(value < 0 || value > 2 ? 2 : (int)value); return VERS_OK; }
  return fail(VERS_ERR_INVALID, std::string("vers_set_option: unknown option ") + name);
}

int32_t vers_mem_stats(uint64_t* out_bytes_now, uint64_t* out_bytes_peak, int32_t reset_peak) {
  dev_mem_stats(out_bytes_now, out_bytes_peak, reset_peak != 0);
  return VERS_OK;
}

int32_t vers_ivf_upload(vers_ivf_t* h, const float* rows, uint64_t n, uint64_t row_stride_bytes, const float* centroids,
                        uint64_t k, uint64_t c_stride_bytes, const uint64_t* assignments) {
  if (!h || (n && (!rows || !assignments)) || (k && !centroids) || row_stride_bytes < (uint64_t)(h ? h->d : 0) * 4 ||
      (k && c_stride_bytes < (uint64_t)h->d * 4) || n > 0xFFFFFFFFull || k > 0xFFFFFFFFull)
    return fail(VERS_ERR_INVALID, "vers_ivf_upload: bad arguments");
  std::vector<uint32_t> a32(n ? n : 1);
  for (uint64_t i = 0; i < n; ++i) {
    if (assignments[i] >= k) return fail(VERS_ERR_INVALID, "vers_ivf_upload: assignment out of range");
    a32[i] = (uint32_t)assignments[i];
  }
  std::unique_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  DevBuf X, A;
  if (int32_t rc = X.reserve((n ? n : 1) * (size_t)h->ldx * sizeof(float))) return rc;
  if (int32_t rc = A.reserve((n ? n : 1) * 4)) return rc;
  if (n) {
    if (h->ldx != h->d) VERS_HIP_TRY(hipMemset(X.p, 0, n * (size_t)h->ldx * sizeof(float)));
    VERS_HIP_TRY(hipMemcpy2D(X.p, (size_t)h->ldx * 4, rows, row_stride_bytes, (size_t)h->d * 4, n, hipMemcpyHostToDevice));
    VERS_HIP_TRY(hipMemcpy(A.p, a32.data(), n * 4, hipMemcpyHostToDevice));
  }
  const size_t cbytes = ((size_t)k * h->ldx ? (size_t)k * h->ldx : 1) * sizeof(float);
  if (int32_t rc = h->centroids.reserve(cbytes)) return rc;
  if (k) {
    VERS_HIP_TRY(hipMemset(h->centroids.p, 0, cbytes));
    VERS_HIP_TRY(hipMemcpy2D(h->centroids.p, (size_t)h->ldx * 4, centroids, c_stride_bytes, (size_t)h->d * 4, k, hipMemcpyHostToDevice));
  }
  return install_index(h, X.as<float>(), h->ldx, n, A.as<uint32_t>(), (uint32_t)k, nullptr);
}

int32_t vers_ivf_add(vers_ivf_t* h, const float* row, uint64_t* out_cluster, uint64_t* out_vec_id) {
  if (!h || !row) return fail(VERS_ERR_INVALID, "vers_ivf_add: bad arguments");
  std::unique_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  VERS_HIP_TRY(hipDeviceSynchronize());  // searches still in flight on any stream read the rows and tables this call changes
  WsLease lease(h);
  if (lease.rc) return lease.rc;
  if (int32_t rc = lease.order_on(nullptr)) return rc;
  if (h->k == 0) return fail(VERS_ERR_EMPTY, "add on an index without centroids (reference: unwrap on None, ivfflat.rs:207)");
  HostStatusSlot slot(h);  // a NaN / spill status latched by an asynchronous _dev search stays there for vers_ivf_poll
  if (h->n_total >= 0xFFFFFFFEull) return fail(VERS_ERR_INVALID, "vec_id space exhausted");
  DevBuf q;
  if (int32_t rc = upload_queries(row, (uint64_t)h->d * 4, 1, h->d, q)) return rc;
  const float* qp = nullptr;
  if (int32_t rc = stage_plain_queries(h, q.as<float>(), h->d, 1, &qp, nullptr)) return rc;
  if (int32_t rc = coarse(h, qp, 1, 1, nullptr)) return rc;  // first-minimum centroid (ivfflat.rs:201-207)
  uint64_t key = 0;
  VERS_HIP_TRY(hipMemcpy(&key, W->probe.p, sizeof(key), hipMemcpyDeviceToHost));
  uint32_t stw = 0;
  VERS_HIP_TRY(hipMemcpy(&stw, W->st_word(), 4, hipMemcpyDeviceToHost));
  if (stw) VERS_HIP_TRY(hipMemset(W->st_word(), 0, 4));
  if ((stw & kStNaN) && h->k >= 2) return fail(VERS_ERR_NAN, "NaN distance in add (reference panics)");
  const uint32_t c = (uint32_t)key;
  const uint32_t vid = (uint32_t)h->n_total;  // the caller's vec_id is ignored, as in the reference (ivfflat.rs:209)
  if (h->h_owner[c] == h->rank) {  // sharded: every rank picks the same list, only its owner stores the row
    if (h->h_len[c] == h->h_cap[c])
      if (int32_t rc = relayout(h)) return rc;
    const uint32_t pos = h->h_off[c] + h->h_len[c];
    hipLaunchKernelGGL(scatter_row_kernel, dim3((h->ld / 4 + 63) / 64), dim3(64), 0, nullptr, qp, h->ld, (uint64_t)pos,
                       h->rows.as<float>());
    VERS_HIP_TRY(hipGetLastError());
    VERS_HIP_TRY(hipMemcpy(h->row_ids.as<uint32_t>() + pos, &vid, 4, hipMemcpyHostToDevice));
    if (int32_t rc = refresh_norms(h, pos, (uint64_t)pos + 1, nullptr)) return rc;
  }
  h->h_len[c] += 1;
  VERS_HIP_TRY(hipMemcpy(h->list_len.as<uint32_t>() + c, &h->h_len[c], 4, hipMemcpyHostToDevice));
  VERS_HIP_TRY(hipMemcpy(h->slot_len.as<uint32_t>() + h->h_slot[c], &h->h_len[c], 4, hipMemcpyHostToDevice));
  h->max_len = std::max(h->max_len, h->h_len[c]);
  h->n_total += 1;
  if (out_cluster) *out_cluster = c;
  if (out_vec_id) *out_vec_id = vid;
  return VERS_OK;
}

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

int32_t vers_shard_plan(const uint64_t* list_lengths, uint64_t k, uint32_t world, uint8_t* out_owner) {
  if ((k && (!list_lengths || !out_owner)) || world == 0 || world > 255) return fail(VERS_ERR_INVALID, "vers_shard_plan: bad arguments");
  shard_plan(list_lengths, k, world, out_owner);
  return VERS_OK;
}

int32_t vers_ivf_set_shard(vers_ivf_t* h, uint32_t rank, uint32_t world) {
  if (!h || world == 0 || world > 255 || rank >= world) return fail(VERS_ERR_INVALID, "vers_ivf_set_shard: bad arguments");
  std::unique_lock<std::shared_mutex> lk(h->index);
  if (h->k != 0) return fail(VERS_ERR_INVALID, "vers_ivf_set_shard: call before build / upload");
  h->rank = rank;
  h->world = world;
  return VERS_OK;
}

int32_t vers_ivf_owners(vers_ivf_t* h, uint8_t* out_owner) {
  if (!h || (h->k && !out_owner)) return fail(VERS_ERR_INVALID, "bad arguments");
  for (uint32_t c = 0; c < h->k; ++c) out_owner[c] = h->h_owner[c];
  return VERS_OK;
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

int32_t vers_ivf_poll(vers_ivf_t* h, void* stream) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  DeviceGuard g(h->device);
  const int32_t rc = sync_status(h, (hipStream_t)stream);
  return rc == kRetrySpill ? VERS_ERR_INVALID : rc;
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
    if (io.direct) {  // (nothing of this workspace is in flight: the previous call / attempt ended with a synchronisation)
      std::memset((char*)io.ids_dev - io.ids_off, 0, io.out_bytes);
      W->st_host = (uint32_t*)((char*)io.ids_dev - io.ids_off + io.st_off);
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

int32_t vers_ivf_shadow_state(vers_ivf_t* h, int32_t* out_active, uint64_t* out_bytes) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  if (out_active) *out_active = (shadow_mode() != 0 && h->shadow_valid && h->rows_bf.p != nullptr && !h->shadow_off) ? 1 : 0;
  if (out_bytes) *out_bytes = h->rows_bf.p ? (uint64_t)h->rows_bf.cap : 0;
  return VERS_OK;
}

int32_t vers_ivf_test_poison_slack(vers_ivf_t* h, float value) {
  if (!h) return fail(VERS_ERR_INVALID, "null handle");
  std::unique_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  VERS_HIP_TRY(hipDeviceSynchronize());
  if (h->cap_rows == 0) return VERS_OK;
  const uint64_t work = h->cap_rows * (h->ld / 4);
  hipLaunchKernelGGL(poison_slack_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, nullptr, h->rows.as<float>(), h->ld,
                     (const uint32_t*)h->row_ids.as<uint32_t>(), h->cap_rows, value);
  VERS_HIP_TRY(hipGetLastError());
  if (int32_t rc = refresh_norms(h, 0, h->cap_rows, nullptr)) return rc;
  VERS_HIP_TRY(hipDeviceSynchronize());
  return VERS_OK;
}

int32_t vers_ivf_test_last_vals(vers_ivf_t* h, uint32_t q, uint64_t* out_vec_ids, float* out_vals, double* out_bound, uint32_t cap, uint32_t* out_n,
                                double* out_info) {
  if (!h || !out_n || (cap && (!out_vec_ids || !out_vals || !out_bound))) return fail(VERS_ERR_INVALID, "bad arguments");
  std::shared_lock<std::shared_mutex> lk(h->index);
  UseLastWs use_ws(h);
  if (!use_ws.ok || !W->last_pre.valid) return fail(VERS_ERR_INVALID, "vers_ivf_test_last_vals: the most recent search did not run the matrix-core list scan");
  const auto lp = W->last_pre;
  if (q >= lp.b) return fail(VERS_ERR_INVALID, "vers_ivf_test_last_vals: no such query in the last batch");
  DeviceGuard g(h->device);
  VERS_HIP_TRY(hipDeviceSynchronize());
  const uint32_t P = lp.P, S = lp.S_max, kp = lp.kp;
  const uint64_t n_pj = (uint64_t)lp.b * P;
  std::vector<uint64_t> keys((size_t)P * S * kp);
  std::vector<uint32_t> pl(P), pp(P), pn(P);
  const uint32_t* pj = W->pj.as<uint32_t>();
  VERS_HIP_TRY(hipMemcpy(keys.data(), W->partials.as<uint64_t>() + (uint64_t)q * P * S * kp, keys.size() * 8, hipMemcpyDeviceToHost));
  VERS_HIP_TRY(hipMemcpy(pl.data(), pj + (uint64_t)q * P, P * 4, hipMemcpyDeviceToHost));
  VERS_HIP_TRY(hipMemcpy(pp.data(), pj + n_pj + (uint64_t)q * P, P * 4, hipMemcpyDeviceToHost));
  VERS_HIP_TRY(hipMemcpy(pn.data(), pj + 3 * n_pj + lp.b + (uint64_t)q * P, P * 4, hipMemcpyDeviceToHost));
  std::vector<float> qrow(h->ldq);
  VERS_HIP_TRY(hipMemcpy(qrow.data(), lp.qp + (uint64_t)q * h->ldq, (size_t)h->ldq * 4, hipMemcpyDeviceToHost));
  uint32_t misc[4] = {0, 0, 0, 0};
  VERS_HIP_TRY(hipMemcpy(misc, h->pre_misc.p, 16, hipMemcpyDeviceToHost));
  double qn = 0.0;
  for (uint32_t j = 0; j < h->ldq; ++j) qn += (double)qrow[j] * (double)qrow[j];
  float xmax2, r2;
  memcpy(&xmax2, &misc[0], 4); memcpy(&r2, &misc[2], 4);
  const PreBound pb = pre_bound(qn, (double)xmax2, lp.shadow ? (double)r2 : 0.0, h->ld, h->metric, lp.shadow);
  if (out_info) { out_info[0] = qn; out_info[1] = xmax2; out_info[2] = lp.shadow ? r2 : 0.0; out_info[3] = pb.global; out_info[4] = pb.common; out_info[5] = kp; out_info[6] = lp.shadow; out_info[7] = h->metric; }
  uint32_t n = 0;
  for (uint32_t j = 0; j < P; ++j) {
    if (pl[j] == kNoList) continue;
    uint32_t off_j = 0;
    VERS_HIP_TRY(hipMemcpy(&off_j, h->slot_off.as<uint32_t>() + pl[j], 4, hipMemcpyDeviceToHost));
    for (uint32_t sq = 0; sq < pn[j] && sq < S; ++sq)
      for (uint32_t i = 0; i < kp; ++i) {
        const uint64_t key = keys[((size_t)j * S + sq) * kp + i];
        if (key == kKeyMax) continue;
        if (n < cap) {
          const uint32_t row = off_j + ((uint32_t)key - pp[j]);
          uint32_t vid = 0;
          VERS_HIP_TRY(hipMemcpy(&vid, h->row_ids.as<uint32_t>() + row, 4, hipMemcpyDeviceToHost));
          const uint32_t vb = order_bits_to_f32_bits((uint32_t)(key >> 32));
          float v; memcpy(&v, &vb, 4);
          out_vec_ids[n] = vid; out_vals[n] = v; out_bound[n] = pb.of((double)v);
        }
        ++n;
      }
  }
  *out_n = n;
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
