// ivf_build.hip -- everything that CHANGES the index: build_index (ivfflat.rs:102-136: k-means driver, best of attempts,
// inverted lists), the row-sharded build over processes, upload of the five fields after load_index, add (ivfflat.rs:200-213).
//
// Device layout (HBM): the reference keeps `values` in vec_id order and gathers list members
// through `ids` (ivfflat.rs:172-174).  Here rows are stored CLUSTER-MAJOR: list c occupies the
// contiguous rows [list_off[c], list_off[c]+list_len[c]) in the reference's list order
// (ascending vec_id for built rows, append order for added ones), followed by slack for `add`;
// row_ids[] maps a storage row back to its vec_id.  Lists start on 64-row boundaries and the rows
// themselves are held in lane-transposed 64-row tiles (scan.hip.h), so a list scan is one linear HBM
// stream of contiguous 1 KiB wave loads.
#include <chrono>
#include <thread>

#include "gemm.hip.h"
#include "ivf_handle.hpp"
#include "prescan.hip.h"

namespace vers {

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

// vers_ivf_add: the ranked list and the stream's status word in one place (one copy back), the word cleared; and the call's three
// table updates (vec id of the new storage row, the list's length by centroid and by slot)
__global__ void add_fetch_kernel(const uint64_t* probe, uint32_t* st_word, uint64_t* out) {
  out[0] = probe[0];
  out[1] = *st_word;
  *st_word = 0u;
}
__global__ void add_tables_kernel(uint32_t* row_id, uint32_t vid, uint32_t* list_len, uint32_t* slot_len, uint32_t len) {
  if (row_id != nullptr) *row_id = vid;
  *list_len = len;
  *slot_len = len;
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

// device-resident `assignments` (usize = u64 in the reference) -> the u32 the kernels use; *bad != 0: one of them is >= k
__global__ void u64_to_u32_checked_kernel(const uint64_t* in, uint64_t n, uint64_t k, uint32_t* out, uint32_t* bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t v = in[i];
  if (v >= k) *bad = 1u;
  out[i] = (uint32_t)v;
}

__global__ void u32_to_u64_kernel(const uint32_t* in, uint64_t n, uint64_t* out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}


// ---- streamed upload (vers_ivf_upload_begin / _chunk / _end): chunk placement -------------------------------------------
// One BLOCK per 64 consecutive positions of a chunk's cluster-sorted order: the source rows are read as they lie (3 KB
// contiguous each), turned through LDS 64 float4 columns at a time (as gather_tiles_kernel), and written into the
// lane-transposed tiles of their lists -- consecutive sorted positions of one list are consecutive storage rows, so the 64
// rows' float4s of a column are runs of contiguous 16-byte pieces.  Row p of the sorted order (source row id = sorted_ids[p],
// list c) goes to position fill[c] + (p - starts[c]) of list c: the chunks arrive in ascending vec id, so a list fills in
// the reference's order (ivfflat.rs:123-127).  Rows of lists another rank owns are skipped; a row past its list's announced
// length is skipped too (advance_fill_kernel reports it).  vec id = ids32[id] (host path: only the owned rows were sent) or
// first + id.
__global__ __launch_bounds__(256) void place_chunk_kernel(const float* X, uint32_t ldx, uint32_t d, uint32_t ld, const uint32_t* sorted_ids,
                                                          const uint32_t* assign, const uint32_t* starts, const uint32_t* list_off,
                                                          const uint32_t* list_len, const uint32_t* fill, const uint8_t* owner, uint32_t rank,
                                                          const uint32_t* ids32, uint32_t first, uint32_t n, float* rows, uint32_t* row_ids) {
  extern __shared__ __attribute__((aligned(16))) f32x4 tl[];  // [64][kGatherCols4 + 1]
  __shared__ uint32_t s_src[kWave], s_dst[kWave];
  __shared__ uint32_t s_any;
  if (threadIdx.x == 0) s_any = 0u;
  __syncthreads();
  if (threadIdx.x < 64) {
    const uint32_t p = blockIdx.x * 64u + threadIdx.x;
    uint32_t src = 0xFFFFFFFFu, dst = 0u;
    if (p < n) {
      const uint32_t id = sorted_ids[p], c = assign[id];
      if (owner == nullptr || owner[c] == rank) {
        const uint32_t pos = fill[c] + (p - starts[c]);
        if (pos < list_len[c]) {
          src = id;
          dst = list_off[c] + pos;
          row_ids[dst] = ids32 ? ids32[id] : first + id;
          s_any = 1u;
        }
      }
    }
    s_src[threadIdx.x] = src;
    s_dst[threadIdx.x] = dst;
  }
  __syncthreads();
  if (!s_any) return;
  const uint32_t ld4 = ld / 4, ldx4 = ldx / 4;
  constexpr uint32_t kPitch = kGatherCols4 + 1;
  for (uint32_t c0 = 0; c0 < ld4; c0 += kGatherCols4) {
    const uint32_t nc = ld4 - c0 < kGatherCols4 ? ld4 - c0 : kGatherCols4;
    for (uint32_t i = threadIdx.x; i < 64u * kGatherCols4; i += 256u) {  // a row's float4s by consecutive threads
      const uint32_t r = i / kGatherCols4, j = i % kGatherCols4, c4 = c0 + j;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      const uint32_t id = s_src[r];
      if (j < nc && id != 0xFFFFFFFFu && c4 < ldx4 && c4 * 4 < d) {
        v = reinterpret_cast<const f32x4*>(X + (uint64_t)id * ldx)[c4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (c4 * 4 + u >= d) v[u] = 0.0f;  // (columns >= d of X are the caller's padding and may hold anything)
      }
      tl[r * kPitch + j] = v;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 64u * nc; i += 256u) {  // a column's 64 rows by consecutive threads
      const uint32_t j = i / 64u, r = i % 64u;
      if (s_src[r] != 0xFFFFFFFFu) *reinterpret_cast<f32x4*>(rows + blocked_index((uint64_t)s_dst[r], (c0 + j) * 4, ld)) = tl[r * kPitch + j];
    }
    __syncthreads();
  }
}

// after a chunk is placed: fill[c] += its rows of list c (owned lists), seen[c] += them (every list, when the chunk held
// ALL its rows); *bad |= 2 when a list received more rows than begin announced
__global__ void advance_fill_kernel(const uint32_t* counts, uint32_t k, const uint8_t* owner, uint32_t rank, const uint32_t* list_len, uint32_t* fill,
                                    uint32_t* seen, uint32_t* bad) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= k) return;
  const uint32_t cnt = counts[c];
  if (seen) seen[c] += cnt;
  if (owner == nullptr || owner[c] == rank) {
    const uint64_t f = (uint64_t)fill[c] + cnt;
    if (f > list_len[c]) { *bad |= 2u; fill[c] = list_len[c]; }
    else fill[c] = (uint32_t)f;
  }
}

}  // namespace vers

namespace vers {
namespace ivf {

// test hook: every storage row that holds no vector (slack behind the lists, tile padding) gets `value` in all its columns,
// then the derived arrays (|x|^2, fp16 shadow, residual) are rebuilt -- what uninitialised device memory may look like
static __global__ void poison_slack_kernel(float* rows, uint32_t ld, const uint32_t* row_ids, uint64_t n_rows, float value) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t r = t / (ld / 4);
  const uint32_t j = (uint32_t)(t % (ld / 4));
  if (r >= n_rows || row_ids[r] != 0xFFFFFFFFu) return;
  reinterpret_cast<f32x4*>(rows + (r >> 6) * 64ull * ld)[(uint64_t)j * 64 + (r & 63)] = f32x4{value, value, value, value};
}
int32_t poison_slack(vers_ivf* h, float value, hipStream_t st) {
  if (h->cap_rows == 0) return VERS_OK;
  const uint64_t work = h->cap_rows * (h->ld / 4);
  hipLaunchKernelGGL(poison_slack_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, h->rows.as<float>(), h->ld,
                     (const uint32_t*)h->row_ids.as<uint32_t>(), h->cap_rows, value);
  VERS_HIP_TRY(hipGetLastError());
  return VERS_OK;
}

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
    // option "rowmajor": -1 (default) = whenever it fits (vers_ivf::rows_rm) unless option "memory" is 1 (compact: ONE f32 copy of the
    // rows -- the tiles; the exact finish then gathers its survivors from them in 16-byte pieces), 0 never, 1 always
    const int rm_opt = (int)opt_get("rowmajor", -1);
    const int rm_mode = rm_opt < 0 && opt_get("memory", 0) == 1 ? 0 : rm_opt;
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
// host wall clock of a phase of build_index, added to BuildStats::*field on scope exit (vers_build_phases)
struct PhaseClock {
  double BuildStats::*field;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit PhaseClock(double BuildStats::*f) : field(f) {}
  ~PhaseClock() { build_stats_add(field, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
};
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
    // slack for `add` behind the list: 1/16 of its length, at least 8 rows, then up to the tile boundary the next list starts on.
    // (Rounds 1-5 kept at least 64: at k = 65536 over 6.25M rows -- lists of ~95 rows -- that alone doubled the storage; a list
    // that outgrows its slack is re-laid-out with 1/8 of head-room, relayout().)
    const uint32_t cap = mine ? round_up(len + std::max<uint32_t>(8u, len / 16u), 64u) : 0u;
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
  VERS_HIP_TRY(hipStreamSynchronize(st));  // (the row placement queued ahead belongs to the install phase's clock)
  PhaseClock derive_clock(&BuildStats::derive_ms);
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
  {  // diagnosis (option "poison_slack_bits"): every new index starts with that value in the rows that hold no vector
    const int64_t poison = opt_get("poison_slack_bits", -1);  // (the f32 bit pattern: 0x7fc00000 NaN, 0x7f800000 inf, ...)
    if (poison >= 0 && h->cap_rows) {
      const uint32_t pbits = (uint32_t)poison;
      float v;
      std::memcpy(&v, &pbits, sizeof(v));
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
    if (h->cap_rows >= 64) {  // whole destination tiles through LDS; (the float4-wise placement of round 1 is kept for indexes of less than one tile)
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
  if (h->cap_rows >= 64 && !segs.empty()) {
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
    PhaseClock alloc_clock(&BuildStats::alloc_ms);
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

// any other way of (re)making the index abandons a streamed upload in progress (vers_ivf_upload_begin .. _end)
static void upload_abandon(vers_ivf* h) {
  if (h->up.open) h->up.close();
}

int32_t build_common(vers_ivf* h, const float* X, uint32_t ldx, uint64_t n, const BuildShard& sh_in, uint64_t num_clusters, uint64_t num_attempts,
                     uint64_t max_iterations, const uint64_t* init_indices, float* out_centroids, uint64_t c_stride_bytes,
                     uint64_t* out_assignments, float* out_cost, int32_t* out_kept, uint64_t* out_iterations) {
  const uint32_t k = (uint32_t)num_clusters;
  upload_abandon(h);
  PhaseClock total_clock(&BuildStats::total_ms);
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
  {
    const auto t_in = std::chrono::steady_clock::now();
    const double derive0 = build_stats().derive_ms;
    if (sh.comm != nullptr && sh.world > 1) {
      if (int32_t rc = install_index_sharded(h, X, ldx, n, sh, best_assign.as<uint32_t>(), k, nullptr)) return rc;
    } else {
      if (int32_t rc = install_index(h, X, ldx, n, best_assign.as<uint32_t>(), k, nullptr)) return rc;
    }
    // (install = everything of this step but what finish_index accounted as derive)
    build_stats_add(&BuildStats::install_ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count() - (build_stats().derive_ms - derive0));
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


// ---- streamed upload: begin / chunk / end (see vers_hip.h) -----------------------------------------------------------------

static int32_t upload_begin_inner(vers_ivf* h, const float* centroids, uint64_t k64, uint64_t c_stride_bytes, const uint64_t* list_lengths, uint64_t n_total);
int32_t upload_begin_locked(vers_ivf* h, const float* centroids, uint64_t k64, uint64_t c_stride_bytes, const uint64_t* list_lengths,
                            uint64_t n_total) {
  const int32_t rc = upload_begin_inner(h, centroids, k64, c_stride_bytes, list_lengths, n_total);
  if (rc) {  // (a failed begin leaves an EMPTY handle: no index, no stored rows for the exhaustive scan, no upload in progress)
    h->k = 0; h->n_total = 0; h->cap_rows = 0;
    upload_abandon(h);
  }
  return rc;
}
static int32_t upload_begin_inner(vers_ivf* h, const float* centroids, uint64_t k64, uint64_t c_stride_bytes, const uint64_t* list_lengths,
                                  uint64_t n_total) {
  const uint32_t k = (uint32_t)k64;
  VERS_HIP_TRY(hipDeviceSynchronize());  // searches still in flight read the storage this call re-plans
  upload_abandon(h);
  h->k = 0;  // no index until _end
  h->n_total = 0;
  std::vector<uint32_t> lens(k ? k : 1, 0);
  uint64_t sum = 0;
  for (uint32_t c = 0; c < k; ++c) {
    if (list_lengths[c] > 0xFFFFFFFFull) return fail(VERS_ERR_INVALID, "vers_ivf_upload_begin: a list longer than 2^32-1 rows");
    lens[c] = (uint32_t)list_lengths[c];
    sum += list_lengths[c];
  }
  if (sum != n_total) return fail(VERS_ERR_INVALID, "vers_ivf_upload_begin: the list lengths do not add up to n_total");
  const size_t cbytes = ((size_t)k * h->ldx ? (size_t)k * h->ldx : 1) * sizeof(float);
  if (int32_t rc = h->centroids.reserve(cbytes)) return rc;
  if (k) {
    VERS_HIP_TRY(hipMemset(h->centroids.p, 0, cbytes));
    VERS_HIP_TRY(hipMemcpy2D(h->centroids.p, (size_t)h->ldx * 4, centroids, c_stride_bytes, (size_t)h->d * 4, k, hipMemcpyHostToDevice));
  }
  if (int32_t rc = plan_storage(h, lens.data(), k, nullptr)) return rc;
  // slack behind the lists and tile padding are zero rows, as after build_index (gather_tiles_kernel writes whole tiles)
  VERS_HIP_TRY(hipMemsetAsync(h->rows.p, 0, (h->cap_rows ? h->cap_rows : 1) * (size_t)h->ld * sizeof(float), nullptr));
  auto& up = h->up;
  if (int32_t rc = up.fill.reserve(2 * (size_t)(k ? k : 1) * sizeof(uint32_t))) return rc;
  VERS_HIP_TRY(hipMemsetAsync(up.fill.p, 0, 2 * (size_t)(k ? k : 1) * sizeof(uint32_t), nullptr));
  if (int32_t rc = up.bad.reserve(16)) return rc;
  VERS_HIP_TRY(hipMemsetAsync(up.bad.p, 0, 16, nullptr));
  if (int32_t rc = h->km.counts.reserve((2 * (size_t)k + 2) * sizeof(uint32_t))) return rc;
  VERS_HIP_TRY(hipStreamSynchronize(nullptr));
  up.h_seen.assign(k ? k : 1, 0u);
  up.cap_rows = h->cap_rows;  // (utils::search_exhaustive over the stored rows sees none until _end)
  h->cap_rows = 0;
  up.k = k;
  up.n_total = n_total;
  up.seen = 0;
  up.open = true;
  return VERS_OK;
}

// one chunk whose rows are device-resident: a32 = assignments as u32 (range-checked by the caller's kernel: up.bad bit 0),
// ids32 = the rows' vec ids (nullptr: first + row), count_seen = the chunk holds ALL rows of its vec id range
static int32_t upload_place(vers_ivf* h, const float* X, uint32_t ldx, const uint32_t* a32, const uint32_t* ids32, uint32_t first, uint32_t n,
                            bool count_seen, hipStream_t st) {
  auto& up = h->up;
  const uint32_t k = up.k;
  if (n == 0 || k == 0) return VERS_OK;
  if (int32_t rc = up.sorted.reserve((size_t)n * sizeof(uint32_t))) return rc;
  uint32_t* counts = h->km.counts.as<uint32_t>();
  uint32_t* starts = counts + k;
  if (int32_t rc = km_group(a32, n, k, up.sorted.as<uint32_t>(), counts, starts, h->km, st)) return rc;
  const uint8_t* owner = h->world > 1 ? h->owner.as<uint8_t>() : (const uint8_t*)nullptr;
  uint32_t* fill = up.fill.as<uint32_t>();
  const size_t lds = 64 * (size_t)(kGatherCols4 + 1) * sizeof(f32x4);
  if (int32_t rc = scan_prepare_launch(place_chunk_kernel, lds)) return rc;
  hipLaunchKernelGGL(place_chunk_kernel, dim3((n + 63u) / 64u), dim3(256), lds, st, X, ldx, h->d, h->ld, (const uint32_t*)up.sorted.as<uint32_t>(), a32,
                     (const uint32_t*)starts, (const uint32_t*)h->list_off.as<uint32_t>(), (const uint32_t*)h->list_len.as<uint32_t>(),
                     (const uint32_t*)fill, owner, h->rank, ids32, first, n, h->rows.as<float>(), h->row_ids.as<uint32_t>());
  hipLaunchKernelGGL(advance_fill_kernel, dim3((k + 255u) / 256u), dim3(256), 0, st, (const uint32_t*)counts, k, owner, h->rank,
                     (const uint32_t*)h->list_len.as<uint32_t>(), fill, count_seen ? fill + k : (uint32_t*)nullptr, up.bad.as<uint32_t>());
  VERS_HIP_TRY(hipGetLastError());
  return VERS_OK;
}

static int32_t upload_check_bad(vers_ivf* h, const char* who) {
  uint32_t bad = 0;
  VERS_HIP_TRY(hipMemcpy(&bad, h->up.bad.p, 4, hipMemcpyDeviceToHost));  // (synchronises the null stream: the chunk is placed)
  if (bad) {
    upload_abandon(h);
    return fail(VERS_ERR_INVALID, std::string(who) + (bad & 1u ? ": assignment out of range" : ": a list received more rows than vers_ivf_upload_begin announced") +
                                      " (the streamed upload is abandoned)");
  }
  return VERS_OK;
}

static int32_t upload_chunk_args(vers_ivf* h, const char* who, uint64_t first_vec_id, uint64_t n) {
  if (!h->up.open) return fail(VERS_ERR_INVALID, std::string(who) + ": no streamed upload in progress (vers_ivf_upload_begin first)");
  if (first_vec_id != h->up.seen) return fail(VERS_ERR_INVALID, std::string(who) + ": chunks must arrive in ascending contiguous order (first_vec_id != rows received so far)");
  if (n > h->up.n_total - h->up.seen) return fail(VERS_ERR_INVALID, std::string(who) + ": more rows than vers_ivf_upload_begin announced");
  return VERS_OK;
}

int32_t upload_chunk_dev_locked(vers_ivf* h, const float* rows_dev, uint64_t ld_floats, const uint64_t* assignments_dev, uint64_t first_vec_id,
                                uint64_t n) {
  if (int32_t rc = upload_chunk_args(h, "vers_ivf_upload_chunk_dev", first_vec_id, n)) return rc;
  auto& up = h->up;
  if (n) {
    if (int32_t rc = up.a32.reserve(n * sizeof(uint32_t))) return rc;
    hipLaunchKernelGGL(u64_to_u32_checked_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, assignments_dev, n, (uint64_t)up.k,
                       up.a32.as<uint32_t>(), up.bad.as<uint32_t>());
    VERS_HIP_TRY(hipGetLastError());
    if (int32_t rc = upload_place(h, rows_dev, (uint32_t)ld_floats, up.a32.as<uint32_t>(), nullptr, (uint32_t)first_vec_id, (uint32_t)n, true, nullptr)) return rc;
    if (int32_t rc = upload_check_bad(h, "vers_ivf_upload_chunk_dev")) return rc;
  }
  up.seen += n;
  return VERS_OK;
}

// Host rows: sub-chunks through a bounded pinned buffer of TWO halves.  Only the rows of the lists this rank owns are packed and cross
// PCIe (1/W of them on a rank of W), with their vec ids; every row is counted on the host for the final check.  While the copy engine
// and the placement kernel work on one half (an event per half says when they are done with it) the host packs the next sub-chunk
// into the other -- with several threads when it is large -- and the device-side `bad` word is read ONCE per chunk, at its end.
// (Round 5 packed row by row on one thread into ONE buffer and read `bad` back after every sub-chunk: pack, H2D and placement never
// overlapped.)
int32_t upload_chunk_host_locked(vers_ivf* h, const float* rows, uint64_t row_stride_bytes, const uint64_t* assignments, uint64_t first_vec_id,
                                 uint64_t n) {
  if (int32_t rc = upload_chunk_args(h, "vers_ivf_upload_chunk", first_vec_id, n)) return rc;
  auto& up = h->up;
  const uint32_t ldx = h->ldx, d = h->d;
  const size_t row_b = (size_t)ldx * 4;
  const size_t stage_bytes = (size_t)opt_get("upload_stage_mb", 256) << 20;
  const uint64_t sub = std::max<uint64_t>(64, std::min<uint64_t>(n ? n : 1, (stage_bytes / 2) / (row_b + 8)));
  const size_t half_b = (sub * (row_b + 8) + 255) & ~(size_t)255;
  if (up.pin_cap < 2 * half_b) {
    if (up.pin) { (void)hipHostFree(up.pin); up.pin = nullptr; up.pin_cap = 0; }
    VERS_HIP_TRY(hipHostMalloc(&up.pin, 2 * half_b, hipHostMallocDefault));
    up.pin_cap = 2 * half_b;
  }
  if (int32_t rc = up.stage.reserve(2 * sub * row_b)) return rc;
  if (int32_t rc = up.ids.reserve(2 * sub * 4)) return rc;
  if (int32_t rc = up.a32.reserve(2 * sub * 4)) return rc;
  for (auto& e : up.half_free)
    if (!e) VERS_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const bool sharded = h->world > 1;
  std::vector<uint32_t> pos(sub);
  bool used[2] = {false, false};
  uint32_t turn = 0;
  for (uint64_t s0 = 0; s0 < n; s0 += sub) {
    const uint64_t m = std::min<uint64_t>(sub, n - s0);
    const uint32_t hf = turn & 1u;
    float* p_rows = (float*)((char*)up.pin + hf * half_b);
    uint32_t* p_ids = (uint32_t*)((char*)p_rows + sub * row_b);
    uint32_t* p_a = p_ids + sub;
    // (the half is free once the copies and the placement queued from it two sub-chunks ago are done)
    if (used[hf]) VERS_HIP_TRY(hipEventSynchronize(up.half_free[hf]));
    uint32_t packed = 0;
    for (uint64_t i = 0; i < m; ++i) {  // which rows go, where, with which id (serial: the owned rows keep their ascending order)
      const uint64_t a = assignments[s0 + i];
      if (a >= up.k) { (void)hipDeviceSynchronize(); upload_abandon(h); return fail(VERS_ERR_INVALID, "vers_ivf_upload_chunk: assignment out of range (the streamed upload is abandoned)"); }
      up.h_seen[a] += 1;
      if (sharded && h->h_owner[a] != h->rank) { pos[i] = 0xFFFFFFFFu; continue; }
      pos[i] = packed;
      p_ids[packed] = (uint32_t)(first_vec_id + s0 + i);
      p_a[packed] = (uint32_t)a;
      ++packed;
    }
    auto pack = [&](uint64_t i0, uint64_t i1) {
      for (uint64_t i = i0; i < i1; ++i) {
        if (pos[i] == 0xFFFFFFFFu) continue;
        float* dst = p_rows + (size_t)pos[i] * ldx;
        std::memcpy(dst, (const char*)rows + (s0 + i) * row_stride_bytes, (size_t)d * 4);
        for (uint32_t j = d; j < ldx; ++j) dst[j] = 0.0f;
      }
    };
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = (size_t)packed * row_b >= (size_t(8) << 20) ? std::min(8u, hw) : 1u;
    if (nt > 1) {
      std::vector<std::thread> th;
      const uint64_t per = (m + nt - 1) / nt;
      for (unsigned t = 1; t < nt; ++t) th.emplace_back(pack, std::min<uint64_t>(m, t * per), std::min<uint64_t>(m, (t + 1) * per));
      pack(0, std::min<uint64_t>(m, per));
      for (auto& t : th) t.join();
    } else {
      pack(0, m);
    }
    if (packed) {
      float* d_rows = up.stage.as<float>() + (size_t)hf * sub * ldx;
      uint32_t* d_ids = up.ids.as<uint32_t>() + (size_t)hf * sub;
      uint32_t* d_a = up.a32.as<uint32_t>() + (size_t)hf * sub;
      VERS_HIP_TRY(hipMemcpyAsync(d_rows, p_rows, (size_t)packed * row_b, hipMemcpyHostToDevice, nullptr));
      VERS_HIP_TRY(hipMemcpyAsync(d_ids, p_ids, (size_t)packed * 4, hipMemcpyHostToDevice, nullptr));
      VERS_HIP_TRY(hipMemcpyAsync(d_a, p_a, (size_t)packed * 4, hipMemcpyHostToDevice, nullptr));
      if (int32_t rc = upload_place(h, d_rows, ldx, d_a, d_ids, 0u, packed, false, nullptr)) return rc;
      VERS_HIP_TRY(hipEventRecord(up.half_free[hf], nullptr));
      used[hf] = true;
    }
    ++turn;
  }
  // once per chunk (synchronises the null stream: everything is placed, both halves are free); a list that received too many rows
  // would also be caught by the per-list counts checked in _end
  if (n)
    if (int32_t rc = upload_check_bad(h, "vers_ivf_upload_chunk")) return rc;
  up.seen += n;
  return VERS_OK;
}

int32_t upload_end_locked(vers_ivf* h) {
  auto& up = h->up;
  if (!up.open) return fail(VERS_ERR_INVALID, "vers_ivf_upload_end: no streamed upload in progress");
  const uint32_t k = up.k;
  const uint64_t n_total = up.n_total;
  auto bail = [&](const std::string& m) { upload_abandon(h); return fail(VERS_ERR_INVALID, "vers_ivf_upload_end: " + m + " (the handle stays empty)"); };
  if (up.seen != n_total) return bail("fewer rows arrived than vers_ivf_upload_begin announced");
  std::vector<uint32_t> f(2 * (size_t)(k ? k : 1), 0u);
  VERS_HIP_TRY(hipMemcpy(f.data(), up.fill.p, f.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  for (uint32_t c = 0; c < k; ++c) {
    if ((uint64_t)f[k + c] + up.h_seen[c] != h->h_len[c]) return bail("list " + std::to_string(c) + " received a different number of rows than announced");
    if (h->h_owner[c] == h->rank && f[c] != h->h_len[c]) return bail("owned list " + std::to_string(c) + " is not complete");
  }
  h->cap_rows = up.cap_rows;
  up.close();
  return finish_index(h, k, n_total, nullptr);
}

}  // namespace ivf
}  // namespace vers

extern "C" {

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

int32_t vers_ivf_upload(vers_ivf_t* h, const float* rows, uint64_t n, uint64_t row_stride_bytes, const float* centroids,
                        uint64_t k, uint64_t c_stride_bytes, const uint64_t* assignments) {
  if (!h || (n && (!rows || !assignments)) || (k && !centroids) || row_stride_bytes < (uint64_t)(h ? h->d : 0) * 4 ||
      (k && c_stride_bytes < (uint64_t)h->d * 4) || n > 0xFFFFFFFFull || k > 0xFFFFFFFFull)
    return fail(VERS_ERR_INVALID, "vers_ivf_upload: bad arguments");
  // = the streamed sequence in one call: list lengths from the assignments, one chunk (staged through a bounded buffer), end
  std::vector<uint64_t> lens(k ? k : 1, 0);
  for (uint64_t i = 0; i < n; ++i) {
    if (assignments[i] >= k) return fail(VERS_ERR_INVALID, "vers_ivf_upload: assignment out of range");
    lens[assignments[i]] += 1;
  }
  std::unique_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  if (int32_t rc = upload_begin_locked(h, centroids, k, c_stride_bytes, lens.data(), n)) return rc;
  if (int32_t rc = upload_chunk_host_locked(h, rows, row_stride_bytes, assignments, 0, n)) return rc;
  return upload_end_locked(h);
}

int32_t vers_ivf_upload_begin(vers_ivf_t* h, const float* centroids, uint64_t k, uint64_t c_stride_bytes, const uint64_t* list_lengths,
                              uint64_t n_total) {
  if (!h || (k && (!centroids || !list_lengths)) || (k && c_stride_bytes < (uint64_t)h->d * 4) || n_total > 0xFFFFFFFFull || k > 0xFFFFFFFFull ||
      (k == 0 && n_total != 0))
    return fail(VERS_ERR_INVALID, "vers_ivf_upload_begin: bad arguments");
  std::unique_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  return upload_begin_locked(h, centroids, k, c_stride_bytes, list_lengths, n_total);
}

int32_t vers_ivf_upload_chunk(vers_ivf_t* h, const float* rows, uint64_t row_stride_bytes, const uint64_t* assignments, uint64_t first_vec_id,
                              uint64_t n) {
  if (!h || (n && (!rows || !assignments)) || row_stride_bytes < (uint64_t)(h ? h->d : 0) * 4)
    return fail(VERS_ERR_INVALID, "vers_ivf_upload_chunk: bad arguments");
  std::unique_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  return upload_chunk_host_locked(h, rows, row_stride_bytes, assignments, first_vec_id, n);
}

int32_t vers_ivf_upload_chunk_dev(vers_ivf_t* h, const float* rows_dev, uint64_t ld_floats, const uint64_t* assignments_dev, uint64_t first_vec_id,
                                  uint64_t n) {
  if (!h || (n && (!rows_dev || !assignments_dev)) || ld_floats < (uint64_t)(h ? h->d : 0) || ld_floats % 4 || ld_floats > 0x3FFFFFFFull)
    return fail(VERS_ERR_INVALID, "vers_ivf_upload_chunk_dev: bad arguments (ld_floats must be >= d and a multiple of 4)");
  std::unique_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  return upload_chunk_dev_locked(h, rows_dev, ld_floats, assignments_dev, first_vec_id, n);
}

int32_t vers_ivf_upload_end(vers_ivf_t* h) {
  if (!h) return fail(VERS_ERR_INVALID, "vers_ivf_upload_end: null handle");
  std::unique_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  return upload_end_locked(h);
}

int32_t vers_ivf_upload_dev(vers_ivf_t* h, const float* rows_dev, uint64_t n, uint64_t ld_floats, const float* centroids_dev, uint64_t k,
                            uint64_t c_ld_floats, const uint64_t* assignments_dev) {
  if (!h || (n && (!rows_dev || !assignments_dev)) || (k && !centroids_dev) || ld_floats < (uint64_t)(h ? h->d : 0) || ld_floats % 4 ||
      ld_floats > 0x3FFFFFFFull || (k && c_ld_floats < (uint64_t)h->d) || n > 0xFFFFFFFFull || k > 0xFFFFFFFFull)
    return fail(VERS_ERR_INVALID, "vers_ivf_upload_dev: bad arguments (ld_floats must be >= d and a multiple of 4)");
  std::unique_lock<std::shared_mutex> lk(h->index);
  DeviceGuard g(h->device);
  upload_abandon(h);
  DevBuf A, bad;
  if (int32_t rc = A.reserve((n ? n : 1) * 4)) return rc;
  if (int32_t rc = bad.reserve(16)) return rc;
  VERS_HIP_TRY(hipMemset(bad.p, 0, 16));
  if (n) {
    hipLaunchKernelGGL(u64_to_u32_checked_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, assignments_dev, n, k, A.as<uint32_t>(),
                       bad.as<uint32_t>());
    VERS_HIP_TRY(hipGetLastError());
  }
  uint32_t any_bad = 0;
  VERS_HIP_TRY(hipMemcpy(&any_bad, bad.p, 4, hipMemcpyDeviceToHost));
  if (any_bad) return fail(VERS_ERR_INVALID, "vers_ivf_upload_dev: assignment out of range");
  const size_t cbytes = ((size_t)k * h->ldx ? (size_t)k * h->ldx : 1) * sizeof(float);
  if (int32_t rc = h->centroids.reserve(cbytes)) return rc;
  if (k) {
    VERS_HIP_TRY(hipMemset(h->centroids.p, 0, cbytes));
    VERS_HIP_TRY(hipMemcpy2D(h->centroids.p, (size_t)h->ldx * 4, centroids_dev, (size_t)c_ld_floats * 4, (size_t)h->d * 4, k, hipMemcpyDeviceToDevice));
  }
  return install_index(h, rows_dev, (uint32_t)ld_floats, n, A.as<uint32_t>(), (uint32_t)k, nullptr);
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
  // (round 5: no allocation and three synchronisations fewer per call -- the row goes through the workspace's upload buffer instead of
  // a buffer allocated and freed per call, the ranked list and the status word come back in ONE copy, the three 4-byte table
  // updates are one small kernel instead of three synchronous copies: 196 -> ~100 us per vector at cfg3's geometry)
  const size_t back_off = ((size_t)h->d * sizeof(float) + 7) & ~(size_t)7;
  if (int32_t rc = W->io_q.reserve(back_off + 16)) return rc;
  VERS_HIP_TRY(hipMemcpyAsync(W->io_q.p, row, (size_t)h->d * sizeof(float), hipMemcpyHostToDevice, nullptr));
  const float* qp = nullptr;
  if (int32_t rc = stage_plain_queries(h, W->io_q.as<float>(), h->d, 1, &qp, nullptr)) return rc;
  if (int32_t rc = coarse(h, qp, 1, 1, nullptr)) return rc;  // first-minimum centroid (ivfflat.rs:201-207)
  uint64_t* const back = reinterpret_cast<uint64_t*>(W->io_q.as<char>() + back_off);  // [0] ranked list | [1] status word
  hipLaunchKernelGGL(add_fetch_kernel, dim3(1), dim3(1), 0, nullptr, (const uint64_t*)W->probe.p, W->st_word(), back);
  VERS_HIP_TRY(hipGetLastError());
  uint64_t got[2] = {0, 0};
  VERS_HIP_TRY(hipMemcpy(got, back, sizeof(got), hipMemcpyDeviceToHost));
  const uint64_t key = got[0];
  const uint32_t stw = (uint32_t)got[1];  // (add_fetch_kernel cleared the word)
  if ((stw & kStNaN) && h->k >= 2) return fail(VERS_ERR_NAN, "NaN distance in add (reference panics)");
  const uint32_t c = (uint32_t)key;
  const uint32_t vid = (uint32_t)h->n_total;  // the caller's vec_id is ignored, as in the reference (ivfflat.rs:209)
  const bool mine = h->h_owner[c] == h->rank;  // sharded: every rank picks the same list, only its owner stores the row
  uint32_t pos = 0;
  if (mine) {
    if (h->h_len[c] == h->h_cap[c])
      if (int32_t rc = relayout(h)) return rc;
    pos = h->h_off[c] + h->h_len[c];
    hipLaunchKernelGGL(scatter_row_kernel, dim3((h->ld / 4 + 63) / 64), dim3(64), 0, nullptr, qp, h->ld, (uint64_t)pos,
                       h->rows.as<float>());
    VERS_HIP_TRY(hipGetLastError());
  }
  h->h_len[c] += 1;
  hipLaunchKernelGGL(add_tables_kernel, dim3(1), dim3(1), 0, nullptr, mine ? h->row_ids.as<uint32_t>() + pos : (uint32_t*)nullptr, vid,
                     h->list_len.as<uint32_t>() + c, h->slot_len.as<uint32_t>() + h->h_slot[c], h->h_len[c]);
  VERS_HIP_TRY(hipGetLastError());
  if (mine)
    if (int32_t rc = refresh_norms(h, pos, (uint64_t)pos + 1, nullptr)) return rc;
  VERS_HIP_TRY(hipStreamSynchronize(nullptr));  // the index is consistent when the call returns (searches on any stream may follow)
  h->max_len = std::max(h->max_len, h->h_len[c]);
  h->n_total += 1;
  if (out_cluster) *out_cluster = c;
  if (out_vec_id) *out_vec_id = vid;
  return VERS_OK;
}

}  // extern "C"
