// ivf_src.hip.h -- how the scan engine (scan.hip.h) sees the index: work-item sources of the coarse quantiser / exhaustive scan
// (SegSrc), the inverted-list scans (IvfSrc) and the single query's item records (Item1Rec / Rec1Src), + the launch of a
// segment scan, which the planner (coarse quantiser) and the search (exhaustive scan) share.
#pragma once
#include "ivf_handle.hpp"
#include "scan.hip.h"

namespace vers {

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

}  // namespace vers

namespace vers {
namespace ivf {
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

}  // namespace ivf
}  // namespace vers
