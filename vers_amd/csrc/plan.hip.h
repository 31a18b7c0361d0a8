// plan.hip.h -- planning of a batch of searches on the device, WITHOUT any cross-block waiting.
//
// search_approximate (ivfflat.rs:153-198) ranks the lists of a query (coarse quantiser) and then walks them
// (ivfflat.rs:166-195).  Between the two the batch has to be re-grouped BY LIST so that a list is streamed once for all
// the queries that probe it.  That is three dependent steps:
//   (1) per query: which lists are scanned, sequence bases, the reference's spill plan      -> plan_query (a WAVE per query)
//   (2) per list : how many (query, probe) pairs, groups and work items; exclusive prefix sums over the lists
//   (3) scatter  : pairs grouped by list, item / group descriptors
// Rounds 1-2 ran them in one launch separated by a software grid barrier (spinning on a device counter).  A spinning
// grid needs all its blocks co-resident, which a plain launch does not promise once several streams / an RCCL kernel share
// the CUs ("the hang"): partially resident spinners of two launches could hold exactly the CU slots the
// other's missing blocks need.  Here nothing spins: step (1) runs in the TAIL of the coarse quantiser's selection
// kernel (its lane j holds the key of probe rank j -- plan_query's operand), steps (2)+(3) are ONE ordinary launch in
// which every block computes the (cheap) prefix sums REDUNDANTLY and then serves the lists it owns (list % blocks).
#pragma once
#include "scan.hip.h"

namespace vers {

constexpr uint32_t kNoList = 0xFFFFFFFFu;
constexpr uint32_t kStNaN = 1u, kStInsufficient = 2u, kStSpillTooDeep = 4u;
// sharded search: a rank that failed LOCALLY still takes part in the batch's one all-gather, with a poisoned partial (every key
// kKeyMax, first id word kPoisonId); every rank's merge latches kStPeerFailed in its stream's status word (vers_ivf_poll -> VERS_ERR_COMM)
constexpr uint32_t kStPeerFailed = 8u;
constexpr uint64_t kPoisonId = 0xDEADFA11DEADFA11ull;
constexpr uint32_t kStNotYet = 0xFFFFFFFFu;  // host-pointer single-query call: the pinned status word before the last merge launch has written it
constexpr uint32_t kNoSeg = 0xFFFFFFFFu;  // padding item of a quad

// Row segments of one list.  seg_target == 0: fixed seg_rows.  Otherwise (matrix-core scan) the list is cut into
// 4 * ceil(len / (4 * seg_target)) nearly equal whole-tile segments, so that the four waves of a quad -- which
// share a query block and a barrier -- carry the same load whatever the list length.
__host__ __device__ __forceinline__ uint32_t list_seg_rows(uint32_t len, uint32_t seg_rows, uint32_t seg_target) {
  if (seg_target == 0) return seg_rows;
  uint32_t n_quads = (len + 4 * seg_target - 1) / (4 * seg_target);
  if (n_quads == 0) n_quads = 1;
  const uint32_t per = (len + 4 * n_quads - 1) / (4 * n_quads);
  const uint32_t seg = (per + 63) / 64 * 64;
  return seg ? seg : 64u;
}
struct ItemDesc {
  uint32_t list, group, seg;
};
struct GroupDesc {
  uint32_t pair_start, nq;
};
struct GroupTotals {
  uint32_t n_items, n_groups, n_pairs, pad;
  uint64_t union_rows;     // sum of len over lists probed by at least one query (algorithmic rows)
  uint64_t streamed_rows;  // rows the scan items actually stream (a list is re-read per query group)
};

// Everything step (1) needs.  b == 0: no planning (the selection kernel of a look-ahead batch).
struct PlanQ {
  uint32_t b = 0, P = 0, k_lists = 0, top_k = 0;
  int ref_mode = 0;
  const uint32_t* list_len = nullptr;   // by centroid index (GLOBAL lengths)
  const uint8_t* owner = nullptr;       // nullable: only lists with owner[L] == rank are scanned on this GPU
  uint32_t rank = 0;
  const uint32_t* list_slot = nullptr;  // centroid index -> slot (the tables below are addressed by slot)
  uint32_t *pj_list = nullptr, *pj_pref = nullptr, *pj_take = nullptr, *np = nullptr;
  uint32_t* pj_nq = nullptr;            // nullable (matrix-core scan only)
  uint32_t *cnt = nullptr, *hot = nullptr;  // zeroed before the launch that plans
  uint32_t hot_ranks = 0, seg_rows = 0, seg_target = 0;
  uint32_t* status = nullptr;
  uint32_t zero_words = 0;              // != 0: the zone that starts at cnt is zeroed by the coarse contraction's launch (no memset in front)
};

// One chunk (64 probe ranks) of a query's plan: lane j holds the key of probe rank c0 + j (kKeyMax = none).
// nprobe mode: every probed list is scanned, pj_pref = running row count.  reference mode (ivfflat.rs:166-195): the walk
// of the ranked lists in closed form -- list j is visited while the rows before it do not yet fill top_k and contributes
// take_j = min(len_j, top_k - rows before); out of lists -> the reference panics.  `carry` / `n_visited` link the chunks.
__device__ __forceinline__ void plan_query_chunk(const PlanQ& a, uint32_t q, int lane, uint32_t c0, uint64_t key, uint32_t& carry,
                                                 uint32_t& n_visited) {
  const uint32_t j = c0 + (uint32_t)lane;
  const uint32_t L = (j < a.P && key != kKeyMax) ? (uint32_t)key : kNoList;  // centroid index
  const uint32_t len = L != kNoList ? a.list_len[L] : 0u;
  const uint32_t slot = L != kNoList ? a.list_slot[L] : kNoList;
  const uint32_t own = (a.owner != nullptr && L != kNoList) ? (uint32_t)a.owner[L] : a.rank;  // (with the lengths: one round trip, not two)
  const uint32_t inc = wave_incl_u32(len);
  const uint32_t pref = carry + inc - len;
  carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, kWave - 1);
  const bool visited = L != kNoList && (!a.ref_mode || pref < a.top_k);
  const uint32_t take = !visited ? 0u : (a.ref_mode ? (len < a.top_k - pref ? len : a.top_k - pref) : a.top_k);
  const bool scan = visited && len > 0 && take > 0 && own == a.rank;
  if (j < a.P) {
    a.pj_list[(uint64_t)q * a.P + j] = scan ? slot : kNoList;
    a.pj_pref[(uint64_t)q * a.P + j] = pref;
    a.pj_take[(uint64_t)q * a.P + j] = take;
    if (a.pj_nq) {  // matrix-core scan: one partial slot per quad of segments of a scanned list
      const uint32_t sr = list_seg_rows(len, a.seg_rows, a.seg_target);
      a.pj_nq[(uint64_t)q * a.P + j] = scan ? ((len + sr - 1) / sr + 3) / 4 : 0u;
    }
  }
  if (scan) atomicAdd(&a.cnt[slot], 1u);
  // This query's tightest thresholds come from its nearest list (the group step orders the work: hot lists first).
  // ("Nearest AMONG THE LISTS THIS GPU SCANS" was tried for sharded indexes in round 4 and lost: DESIGN.md Appendix A.)
  if (scan && c0 == 0 && j < a.hot_ranks) a.hot[slot] = 1u;
  n_visited += (uint32_t)__popcll(__ballot(visited));
}
__device__ __forceinline__ void plan_query_finish(const PlanQ& a, uint32_t q, int lane, uint32_t carry, uint32_t n_visited) {
  if (lane == 0) {
    a.np[q] = n_visited;
    if (a.ref_mode && a.top_k > 0 && carry < a.top_k) atomicOr(a.status, a.P >= a.k_lists ? kStInsufficient : kStSpillTooDeep);
  }
}

}  // namespace vers
