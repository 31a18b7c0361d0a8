// ivf_plan.hip -- the first half of search_approximate (ivfflat.rs:155-161 and the plan of the walk, 166-195): the coarse
// quantiser (single query: coarse1_kernel; batches: matrix-core pre-selection + exact re-score, gemm.hip.h; exact scans
// otherwise), the per-query plan (which lists, sequence bases, the reference's spill plan) and the re-grouping of the
// (query, list) pairs BY LIST into work items -- all on the device, no cross-block waiting (plan.hip.h).
#include "gemm.hip.h"
#include "ivf_src.hip.h"
#include "prescan.hip.h"

namespace vers {

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
// (256-thread blocks were tried in round 4 and lost -- the kernel itself is slower: DESIGN.md Appendix A)
constexpr uint32_t kGroupThreads = 1024, kGroupMaxBlocks = 64;
constexpr int kGroupWaves = (int)(kGroupThreads / kWave);
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
  unsigned long long* stamps;  // diagnosis (option "scan_debug" & 16): [16..19] 100 MHz clock at the phase boundaries, block 0
};
// a table entry this block stored itself a phase ago: read past the vector L1 (which may hold the line from before the store)
__device__ __forceinline__ uint32_t ld_l2(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool owns_list(uint32_t L) { return ((L >> 2) % gridDim.x) == blockIdx.x; }

// Four exclusive prefix sums over the lists (pairs, groups, items of hot lists, items of the others) in rounds of 4096
// lists: a thread owns FOUR consecutive lists (three 16-byte loads, a serial scan in registers), the waves scan the
// thread totals by shuffles, 16 wave totals go through LDS, a running carry links the rounds.  One round and two block
// barriers at 4096 lists.
__device__ __forceinline__ uint32_t group_lists(const GroupArgs& a, uint32_t* tab) {
  __shared__ uint32_t wp[kGroupWaves], wg[kGroupWaves], wi[kGroupWaves], wh[kGroupWaves];
  __shared__ unsigned long long ur, sr;
  const uint32_t* cnt = a.cnt; const uint32_t* list_len = a.slot_len; const uint32_t* hot = a.hot;
  const uint32_t k_lists = a.k_lists, QG = a.QG;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (threadIdx.x == 0) { ur = sr = 0; }
  uint32_t cp = 0, cg = 0, ci = 0, ch = 0;  // carries (identical in every thread); ci: other lists' items, ch: hot lists' items
  unsigned long long my_ur = 0, my_sr = 0;
  auto wave_incl = [&](uint32_t x) { return wave_incl_u32(x); };
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
    for (int w = 0; w < kGroupWaves; ++w) {
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
  Item1Rec* recs = nullptr; const uint32_t* list_off = nullptr; uint32_t S_max = 0;  // recs != nullptr: the items as records (scan1_kernel); list_off: storage row of a list BY CENTROID
  uint32_t *pj_nq = nullptr, *qflags = nullptr, *fail_cnt = nullptr;  // the shadow scan's exact finish (scan1h_kernel): records per probe; flags and the queue's count, zeroed here
  u32x4* ff_begin; uint32_t ff_vec16;  // the list scan's partial slots: filled with all ones (empty) by whoever plans
  unsigned long long* stamps = nullptr;  // diagnosis (option "scan_debug" & 16): [48..50] 100 MHz clock after the merge, the list tables, the plan's stores
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
  // (the records' operands by CENTROID index -- list_off[L], and the stored length IS list_len[L] -- so that they are requested
  // together with the slot instead of a round trip behind it: round 3 read slot_off[slot] / slot_len[slot])
  const uint32_t loff = (a.recs && L != kNoList) ? a.list_off[L] : 0u, llen = len;
  const uint32_t own = (a.owner != nullptr && L != kNoList) ? (uint32_t)a.owner[L] : a.rank;  // (with the lengths: one round trip, not two)
  auto excl_scan = [&](uint32_t v) { return wave_incl_u32(v) - v; };  // exclusive prefix sum over the 64 lanes
  const uint32_t pref = excl_scan(len);
  if (a.stamps && lane == 0) a.stamps[49] = __builtin_amdgcn_s_memrealtime();
  const uint32_t total_rows = (uint32_t)__shfl(pref + len, kWave - 1, kWave);
  // reference mode (ivfflat.rs:166-195) in closed form: list j is visited while the rows before it do not yet
  // fill top_k, and contributes take_j = min(len_j, top_k - rows before it)
  const bool visited = L != kNoList && (!a.ref_mode || pref < top_k);
  const uint32_t take = !visited ? 0u : (a.ref_mode ? (len < top_k - pref ? len : top_k - pref) : top_k);
  const bool scan = visited && len > 0 && take > 0 && own == a.rank;
  if (lane < (int)P) {
    a.pj_list[lane] = scan ? slot : kNoList;
    a.pj_pref[lane] = pref;
    a.pj_take[lane] = take;
  }
  if (a.pj_nq != nullptr) {
    if (lane < (int)P) {
      a.pj_nq[lane] = scan ? (len + a.seg_rows - 1) / a.seg_rows : 0u;
      a.qflags[lane] = 0u;
    }
    if (lane == 0) *a.fail_cnt = 0u;
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
  unsigned long long* stamps;  // diagnosis (option "scan_debug" & 16): [32..39] 100 MHz clock at the phase boundaries of the LAST block
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
    wave_rank_sort64(key, lane);  // (unique: they carry their row)
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

}  // namespace vers

namespace vers {
namespace ivf {

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

// batches: MFMA pre-selection + exact re-score + certificate (gemm.hip.h); same output as the exact scan, bit for bit.
// qp: staged queries [round_up(b, kGemmBM)][ldq]; probe_out [b][P].  W->gbuf is the only scratch.
// plan != nullptr: the selection kernel also makes every query's plan (plan.hip.h step 1) in its tail.
int32_t coarse_mfma(vers_ivf* h, const float* qp, uint32_t b, uint32_t P, uint64_t* probe_out, hipStream_t st, const PlanQ* plan) {
  const uint32_t M_pad = round_up(b, kGemmBM);
  const uint32_t PS = std::min<uint32_t>(kMaxTopK, P + 16);
  if (int32_t rc = W->gbuf.reserve((size_t)M_pad * h->k_pad * sizeof(float))) return rc;
  // (event records only when the call is timed at all -- vers_set_option("scan_events") -- and never on the look-ahead stream)
  const bool timed = W->ev_on && (st != W->ahead_stream || W->ahead_stream == nullptr);
  if (timed) VERS_HIP_TRY(hipEventRecord(W->evc[0], st));
  const __bf16* cs = h->centroids_gs.as<__bf16>();
  // the planning tables the selection's tail counts into: zeroed by the bf16x3 contraction's blocks, by a memset otherwise
  const bool zero_in_gemm = plan && plan->zero_words && (gemm_x3_mask() & 2) != 0;
  if (plan && plan->zero_words && !zero_in_gemm) VERS_HIP_TRY(hipMemsetAsync(plan->cnt, 0, (size_t)plan->zero_words * sizeof(uint32_t), st));
  VERS_HIP_TRY(launch_gemm<false>((gemm_x3_mask() & 2) != 0, M_pad / kGemmBM, h->k_pad / kGemmBN, st, qp, h->centroids_g.as<float>(),
                                  h->cnorm.as<float>(), h->ldq, h->k_pad, W->gbuf.as<float>(), h->metric, 0, nullptr, nullptr, nullptr, cs,
                                  cs ? cs + (size_t)h->k_pad * h->ldq : nullptr, zero_in_gemm ? plan->cnt : nullptr, zero_in_gemm ? plan->zero_words : 0u));
  if (timed) VERS_HIP_TRY(hipEventRecord(W->evc[1], st));
  if (P + 16 > (uint32_t)kMaxTopK) {  // more ranked lists than a key per lane: the wide selection (four waves per query, the query in LDS)
    const size_t lds = (size_t)h->ldq * sizeof(float);
    if (int32_t rc = scan_prepare_launch(coarse_select_wide_kernel, lds)) return rc;
    hipLaunchKernelGGL(coarse_select_wide_kernel, dim3(b), dim3(kWave * kSelWideWaves), lds, st, W->gbuf.as<float>(), h->k_pad, h->k,
                       h->centroids_g.as<float>(), h->ldq, qp, h->ldq, h->ldq, coarse_mode() == 2 ? __builtin_inff() : h->cmax2, P, P + 32u,
                       probe_out, W->st_word(), h->coarse_stat.as<uint32_t>(), h->metric, plan ? *plan : PlanQ{}, b);
  } else
  hipLaunchKernelGGL(coarse_select_rescore_kernel, dim3(b), dim3(kWave * kSelWaves), 0, st, W->gbuf.as<float>(), h->k_pad, h->k,
                     h->centroids_g.as<float>(), h->ldq, qp, h->ldq, h->ldq, coarse_mode() == 2 ? __builtin_inff() : h->cmax2, P, PS,
                     probe_out, W->st_word(), h->coarse_stat.as<uint32_t>(), h->metric,
                     (scan_debug_flags() & 16u) && W->stamps.p ? W->stamps.as<unsigned long long>() : (unsigned long long*)nullptr,
                     plan ? *plan : PlanQ{}, b);
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
int32_t coarse(vers_ivf* h, const float* qp, uint32_t b, uint32_t P, hipStream_t st, uint32_t* out_n_segs,
               const PlanQ* plan, bool* planned) {
  // (the contraction reads whole 128-row tiles: the staged block is padded to them, a caller's block used in place is a whole
  // number of them; the selection keeps P + 16 keys: one per lane)
  static_assert(kGemmBM == 128 && kMaxTopK == 64, "coarse_uses_mfma spells these out");
  if (coarse_uses_mfma(h, qp, b, P)) {
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

// Geometry, staging, coarse quantiser and planning of one search call: everything up to the list scan's launch.  On return the
// per-batch tables of the leased workspace are final (in stream order) and `s` says which scan the search runs.
int32_t plan_search(vers_ivf* h, const float* q_dev, uint64_t ldq_in, uint32_t b, uint32_t top_k, uint32_t nprobe, hipStream_t st,
                    SearchPlan& s) {
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
  // ... and the coarse scan with them in coarse1_kernel (option "coarse1" = 0: the ordered-chain scan + plan1_kernel, for A/B runs)
  const bool c1_on = opt_get("coarse1", 1) != 0;
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
  // The ordered-chain kernels stage a group's queries in LDS (QG x ld floats): rows too long for a group of 16 take 8, then one
  // query per item (scalar operands: no LDS at all) -- the reference has no dimension cap (ivfflat.rs:153), neither has this path
  // (round 4 returned "vector dimension too large" at d > 2560 with 16-query groups).
  while (QG > 1 && scan_lds_bytes(QG, h->ld) > 160u * 1024u) QG = QG == 16 ? 8 : 1;
  // Batches in nprobe mode: the list scan runs on the matrix cores with an exact finish (prescan.hip.h); same bits.
  // option "prescan" = 0 keeps the ordered-chain scan, = 2 makes every certificate fail (exercises the exact fallback).
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
  // Results wider than one key per lane leaves room for (top_k + 16 > 64), up to kWideMaxKp - 32 = 200 keys: WIDE candidate lists, four keys
  // per lane through the scan's compactions and the finish (wide.hip.h, finish_wide.hip.h), on the fp16 shadow with hi-only query blocks
  // (round 6; until then the ordered chains, 64 ranks per pass over the f32 rows: 64 k q/s at top_k = 64, 32 k at 100, batch 256, cfg3).
  // Option "wide_k" = 0: the ordered chains (A/B runs).
  constexpr uint32_t kWideSlack = 32;
  const bool wide_k = !ref_mode && b > 1 && use_shadow && top_k + kPreMinSlack > kPreMaxKp && top_k + kWideSlack <= kWideMaxKp && opt_get("wide_k", 1) != 0;
  if (wide_k) kp = top_k + kWideSlack;
  // The block's query operand must fit LDS next to the candidate buffers: 32 queries up to d = 1152; the NARROW variant's 16 up
  // to d = 2304 (d = 1536 -- a dimension the reference's own bindings instantiate, vers-py/src/lib.rs:26-65 -- went to the
  // ordered-chain scan until round 4, ~3x slower).  option "pre_narrow" = 1 forces the narrow blocks (tests, A/B).
  // Round 5: on the fp16 shadow a query block that does not fit with both halves of its fp16 hi + lo split drops the lo half
  // (prescan_kernel_g<.., LO = false>): 32 queries per block up to d = 2304 -- at d = 1536 the 16-query blocks streamed every list
  // probed by more than 16 queries once per extra group, 2.15x the union's bytes -- and 16 up to d = 4608; the certificate charges
  // the query's measured fp16 residual instead (pre_bound).  option "pre_hi_only" = 1 forces it at every d (tests, A/B).
  const bool force_hi = opt_get("pre_hi_only", 0) != 0;
  uint32_t pre_nq = 0;
  bool pre_hi_only = false;
  {
    const bool narrow = knobs().pre_narrow;
    auto fits = [&](uint32_t nq, bool hi) { return prescan_lds_bytes_g(h->ld, kp, nq, hi) <= 160u * 1024u; };
    // 64 queries per block (two sets of 32, hi-only) wherever they fit -- d <= 960 with the default slack --: lists probed by 33 .. 64
    // queries of the batch are then streamed once instead of twice (option "pre_wide" = 0: the 32-query hi + lo blocks of rounds 2-4)
    const bool wide_on = opt_get("pre_wide", 1) != 0;
    if (wide_k) {  // (256-key buffers: 64 KB for 32 queries next to a hi-only query block)
      if (!narrow && fits(kPreQ, true)) { pre_nq = kPreQ; pre_hi_only = true; }
      else if (fits(kPreQNarrow, true)) { pre_nq = kPreQNarrow; pre_hi_only = true; }
    }
    else if (!narrow && wide_on && use_shadow && fits(kPreQWide, true)) { pre_nq = kPreQWide; pre_hi_only = true; }
    else if (!narrow && !(force_hi && use_shadow) && fits(kPreQ, false)) pre_nq = kPreQ;
    else if (!narrow && use_shadow && fits(kPreQ, true)) { pre_nq = kPreQ; pre_hi_only = true; }
    else if (!(force_hi && use_shadow) && fits(kPreQNarrow, false)) pre_nq = kPreQNarrow;
    else if (use_shadow && fits(kPreQNarrow, true)) { pre_nq = kPreQNarrow; pre_hi_only = true; }
  }
  // Small batches too (round 3 required >= 2 queries per list on average and sent batch 8 .. 128 at nlist = 4096 to one
  // ordered-chain scan per (query, list) pair -- every list re-read per query, f32 rows): a list probed by ONE query of the batch
  // is still streamed from the half-size shadow at the chip's rate, and the staging of a mostly empty query block costs less
  // than the bytes it saves.  option "pre_min_batch" (default 4; measured at cfg3: batch 4 171 vs 209 us, batch 2 equal) is the smallest batch that takes this path.
  const bool pre_batch = QG != 1 || (b >= pre_min_batch_ref().load(std::memory_order_relaxed) && b > 1);
  // A single query takes the shadow too (round 5, scan1h_kernel in ivf_search.hip): half the bytes of the ordered-chain scan of the
  // f32 rows, then the same exact finish.  vers_set_option("single_shadow", 0): the ordered-chain scan (A/B runs).
  // (its kernels stage the query in LDS: rows beyond ~36 k columns keep the scalar-operand ordered chains -- no dimension cap on this path either)
  const bool one1_pre = one1 && single_shadow_ref().load(std::memory_order_relaxed) != 0 && use_shadow && !ref_mode && pre_mode != 0 && top_k + kPreMinSlack <= kPreMaxKp &&
                        (size_t)h->ld * sizeof(float) + 24576 <= 160u * 1024u;  // (the finish: the query + 8 KB of exchange area + 14 KB of static buffers)
  // (any nprobe up to kPreMaxP: the work items carry (query, list) pairs; only the RANKING of more than 48 lists leaves the matrix cores
  // -- one key per lane in the selection -- and runs exactly, 64 ranks per pass.  Rounds 1-5 sent nprobe > 64 to the ordered chains.)
  const bool use_pre = one1_pre || (pre_batch && !ref_mode && pre_mode != 0 && (top_k + kPreMinSlack <= kPreMaxKp || wide_k) && P <= kPreMaxP && pre_nq != 0);
  if (use_pre && !one1_pre) QG = (int)pre_nq;
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
  if (use_pre && !one1_pre) seg_rows = (uint32_t)round_up64(std::max<uint64_t>(256, (avg_len_all + 3) / 4), kWave);
  // single query on the shadow: a record is a block's work, a tile per wave -- at most the block's 4 tiles; two blocks fit a CU, so
  // about 1.5 records per CU keeps every record resident at once whatever the probed lists' lengths (cfg3: 87 k rows -> ~350 records)
  if (one1_pre) seg_rows = (uint32_t)std::min<uint64_t>(256, round_up64(std::max<uint64_t>(1, ((uint64_t)P * avg_len_all * 2) / (3 * (uint64_t)h->n_cu)), kWave));
  // (cutting the lists of a SHARDED scan into finer quads -- 2.7 whole-list quads per CU at 8 ranks -- was tried in rounds 3-4 and lost: a
  // short item pays its pipeline fill and its waves' waits for each other whatever the staging costs)
  if (knobs().seg_rows > 0) seg_rows = (uint32_t)round_up64(std::max(64l, knobs().seg_rows), kWave);  // tuning knob
  if (one1_pre && seg_rows > 256) seg_rows = 256;
  // matrix-core scan: per-list balanced segments of about seg_rows rows (list_seg_rows)
  // (same-box A/B at cfg3, round 1: 5.96 ms vs 6.27 ms with fixed 640-row segments)
  const uint32_t seg_target = use_pre && !one1_pre ? seg_rows : 0u;
  // segments per list at most; partial slots per (query, probe): one per segment, or one per QUAD of segments (matrix-core scan)
  const uint32_t S_seg = seg_target ? 4 * std::max<uint32_t>(1, (h->max_len + 4 * seg_target - 1) / (4 * seg_target))
                                    : std::max<uint32_t>(1, (h->max_len + seg_rows - 1) / seg_rows);
  const uint32_t S_max = use_pre && !one1_pre ? (S_seg + 3) / 4 : S_seg;
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
  pq.cnt = cnt; pq.hot = hot; pq.hot_ranks = 1u; pq.seg_rows = seg_rows; pq.seg_target = seg_target;
  pq.status = W->st_word();
  bool planned = false;
  uint32_t n_segs_c = 0;
  if (took) {  // staged queries and ranked lists of this batch were computed ahead (vers_ivf_coarse_ahead_dev)
    if (!one1) VERS_HIP_TRY(hipMemsetAsync(cnt, 0, zero_words * sizeof(uint32_t), st));
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
    // the zeroed zone: by the coarse contraction's own launch when the ranking runs on the matrix cores (coarse_mfma), else here
    const bool zero_in_coarse = !one1 && zero_words <= 0xFFFFFFFFull && coarse_uses_mfma(h, qp, b, P);
    if (zero_in_coarse) pq.zero_words = (uint32_t)zero_words;
    else if (!one1) VERS_HIP_TRY(hipMemsetAsync(cnt, 0, zero_words * sizeof(uint32_t), st));
    if (!one1_fused)
      if (int32_t rc = coarse(h, qp, b, P, st, one1 ? &n_segs_c : nullptr, one1 ? nullptr : &pq, &planned)) return rc;
    probe = W->probe.as<uint64_t>();
  }

  if (n_pass > 1)
    if (int32_t rc = W->lower.reserve(n_pj * sizeof(uint64_t))) return rc;
  if (one1) {
    // (the shadow scan's exact finish reads its slots as one flat array: filled too)
    const bool fill_in_kernel = part_bytes <= (size_t(4) << 20);  // (a block fills a few hundred KB faster than a launch costs)
    if (!fill_in_kernel) VERS_HIP_TRY(hipMemsetAsync(W->partials.p, 0xFF, part_bytes, st));
    Plan1Args pa;
    pa.cpart = W->cpart.as<uint64_t>(); pa.n_segs_c = n_segs_c; pa.P = P; pa.k_lists = k_l; pa.top_k = top_k; pa.ref_mode = ref_mode;
    pa.list_len = h->list_len.as<uint32_t>(); pa.owner = h->world > 1 ? h->owner.as<uint8_t>() : (const uint8_t*)nullptr; pa.rank = h->rank;
    pa.seg_rows = seg_rows; pa.probe = W->probe.as<uint64_t>(); pa.pj_list = pj_list; pa.pj_pref = pj_pref; pa.pj_take = pj_take; pa.np = np;
    pa.cnt = cnt; pa.pair_off = pair_off; pa.group_off = group_off; pa.pairs = W->pairs.as<uint32_t>(); pa.items = W->items.as<ItemDesc>();
    pa.groups = W->groups.as<GroupDesc>(); pa.tot = tot; pa.status = W->st_word(); pa.list_slot = h->list_slot.as<uint32_t>();
    pa.ff_begin = reinterpret_cast<u32x4*>(W->partials.p); pa.ff_vec16 = fill_in_kernel ? (uint32_t)((part_bytes + 15) / 16) : 0u;
    pa.recs = W->items.as<Item1Rec>(); pa.list_off = h->list_off.as<uint32_t>(); pa.S_max = S_max;
    if (one1_pre) { pa.pj_nq = pj_nq; pa.qflags = qflags; pa.fail_cnt = fail_list + b; }
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
  const uint32_t g_blocks = (uint32_t)std::min<uint64_t>(kGroupMaxBlocks, std::max<uint64_t>(1, (n_pj + 2 * kGroupThreads - 1) / (2 * kGroupThreads) + k_l / (kGroupThreads / 4)));
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
  s.P = P; s.ref_mode = ref_mode; s.one1 = one1; s.one1_pre = one1_pre; s.QG = QG; s.use_pre = use_pre; s.use_shadow = use_shadow; s.pre_hi_only = use_pre && pre_hi_only; s.pre_mode = pre_mode;
  s.kp = kp; s.k_keep = k_keep; s.n_pass = n_pass; s.seg_rows = seg_rows; s.seg_target = seg_target; s.S_max = S_max;
  s.items_bound = items_bound; s.part_bytes = part_bytes;
  s.pj_list = pj_list; s.pj_pref = pj_pref; s.pj_take = pj_take; s.np = np; s.pj_nq = pj_nq; s.cnt = cnt; s.pair_off = pair_off;
  s.group_off = group_off; s.quad_ctr = quad_ctr; s.fail_list = fail_list; s.qflags = qflags; s.tot = tot; s.qp = qp; s.took = took;
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

}  // namespace ivf
}  // namespace vers
