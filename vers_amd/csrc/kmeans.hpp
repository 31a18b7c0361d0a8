// kmeans.hpp -- device-level k-means steps shared by kmeans.hip (C ABI primitives) and
// ivf_build.hip (build_index).  All pointers are device pointers; nothing synchronises unless noted.
#pragma once
#include "common.hpp"

namespace vers {

// ---- sort.hip ---------------------------------------------------------------------
size_t group_by_cluster_temp_bytes(uint32_t n, uint32_t k);
int32_t group_by_cluster(const uint32_t* assign, uint32_t n, uint32_t k, uint32_t* sorted_ids, void* temp,
                         size_t temp_bytes, hipStream_t st);

// ---- kmeans.hip ---------------------------------------------------------------------
// Simple owning device buffer (grow-only).
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int32_t reserve(size_t bytes);
  void release();
  template <class T> T* as() const { return (T*)p; }
  ~DevBuf() { release(); }
};

void dev_mem_stats(uint64_t* now, uint64_t* peak, bool reset_peak);  // bytes held through DevBuf, process-wide
void dev_mem_account(int64_t delta);  // for the one allocation made outside DevBuf::reserve

struct KMeansScratch {
  // the assign cascade's verdict for passes over (cascade_n, cascade_k): -1 not probed yet, 1 the <hi, hi> first filter pays, 0 it does not (km_assign_mfma)
  int cascade = -1;
  uint64_t cascade_n = 0;
  uint32_t cascade_k = 0;
  DevBuf cblocked;  // centroids in the scan layout (lane-transposed tiles)
  DevBuf qblocks;   // interleaved point blocks of one assign batch
  DevBuf keys;      // u64 argmin keys of one assign batch
  DevBuf sort_tmp;  // group_by_cluster temp
  DevBuf status;    // u32 status word(s)
  DevBuf counts;    // u32 [k] + starts [k+1]
  DevBuf misc;      // cost scalar, equality flag
  // matrix-core assign (km_assign_mfma)
  DevBuf cg;        // centroids row-major [k_pad][ldq], zero padded
  DevBuf cg_f16;    // the same as fp16 [k_pad][ldq] + 16 bytes: the largest squared fp16 residual of a row (operand of the cascade's single-product filter)
  DevBuf cg_s;      // the same as bf16 hi | lo halves [2][k_pad][ldq] (operand of the bf16x3 contraction, split once per pass)
  DevBuf cnorm;     // |c|^2 [k_pad] (+inf padding) + max at [k_pad]
  DevBuf xp;        // staged point batch [mb][ldq] when X cannot be used in place
  DevBuf xh;        // the point batch as fp16 [mb][ldq], zero padded (operand of dist_gemm_h_kernel)
  DevBuf gt;        // Gt [k_pad][mb]
  DevBuf best;      // u32 [mb] candidate + f32 [mb] second-best value
  DevBuf fb;        // u32 [n] uncertified points + counter at [n]
  DevBuf fbq;       // large k: u32 [n] points queued for the tile re-scan + counter at [n], f32 [n] their G thresholds
  DevBuf xf, fa, fm;  // gathered uncertified points and their exact results
};

// assign_to_clusters (ivfflat.rs:29-46): out_assign[i] = first argmin_c D(X[i], C[c]);
// out_mind[i] (optional) = that minimum distance, bit-exact D(X[i], C[assign[i]]).
// status bit0 is set on a NaN distance.  X [n][ldx], C [k][ldc] row-major (pad columns zero).
// metric 0: squared L2 (the reference); 1: cosine distance 1 - dot (base.rs:153-155) -- the metric extension.
int32_t km_assign(const float* X, uint32_t ldx, uint64_t n, const float* C, uint32_t ldc, uint32_t k, uint32_t d,
                  uint32_t* out_assign, float* out_mind, KMeansScratch& ws, int n_cu, hipStream_t st, int metric);

// Same contract and same bits as km_assign, through the f32 matrix cores: Gt = |c|^2 - 2 C X^T per point batch,
// per-point best / second-best approximate value, exact re-score of the candidate in the reference's arithmetic,
// certificate, and the exact scan (km_assign) for the points that fail it.  Synchronises the stream once.
int32_t km_assign_mfma(const float* X, uint32_t ldx, uint64_t n, const float* C, uint32_t ldc, uint32_t k, uint32_t d,
                       uint32_t* out_assign, float* out_mind, KMeansScratch& ws, int n_cu, hipStream_t st, int metric);
// policy: option "assign" = 1 exact scan, 2 matrix cores always, otherwise by problem size
bool km_use_mfma(uint64_t n, uint32_t k, uint32_t d);
// option "gemm_x3" (bit mask): bit 0 the k-means assign contraction, bit 1 the coarse quantiser's, as three bf16
// products of hi/lo-split operands (gemm.hip.h: dist_gemm_x3_kernel) instead of the f32 MFMA kernel.  Same results.
int gemm_x3_mask();
void set_gemm_x3_mask(int m);  // (vers_set_option: same-process A/B in bench.py)

// Measurement hook (vers_build_stats): where the time of the builds of this process went.  HIP events on the build's stream,
// read at the synchronisation points the build has anyway.
struct BuildStats {
  double gemm_ms = 0;        // assign contraction launches (dist_gemm*_kernel<true>), summed
  double gemm_launches = 0;
  double gemm_flop = 0;      // 2 * points * k * d of those launches (algorithmic: unpadded)
  double assign_ms = 0;      // whole matrix-core assign passes (contraction + arg-min merge + exact re-score + exact re-scans), host wall clock
  double assign_passes = 0;
  double update_ms = 0;      // update_centroids (grouping excluded)
  double cost_ms = 0;        // the one-lane cost fold
  double redone_points = 0;  // points whose certificate failed (settled by the exact kernels)
  // host wall clock of the phases of build_index (vers_build_phases): the build synchronises at every phase boundary anyway
  double alloc_ms = 0;       // the build's device allocations (k-means scratch, assignments, centroid copies)
  double assign_first_ms = 0;  // the FIRST assign pass since the last reset (it pays the cold start: code load, first launches, first touch)
  double install_ms = 0;     // inverted lists: grouping by list, storage plan + allocation, row placement (sharded: + the rows-to-owners exchange)
  double derive_ms = 0;      // what derives from the stored rows and centroids: scan-layout / MFMA operands of the centroids, |x|^2, fp16 shadow, row-major copy
  double total_ms = 0;       // whole build_index calls (entry to return, incl. read-back of centroids / assignments)
};
// Every update of the process-wide statistics goes through these (one mutex: builds of several handles may run side by side).
void build_stats_add(double BuildStats::*field, double v);
BuildStats build_stats();  // (a copy, under the mutex)
// Elapsed time of a stretch of a stream, added to BuildStats::*field when the stream next synchronises (km_timers_collect).
// RAII: the stretch belongs to the scope that opened it -- it is closed on every path out of it (an early error return
// used to leave it open and its two events leaked), and two builds on two threads cannot close each other's stretches.
struct KmTimer {
  hipStream_t st;
  double BuildStats::*field;
  hipEvent_t a = nullptr, b = nullptr;
  KmTimer(hipStream_t stream, double BuildStats::*f);
  ~KmTimer();
  KmTimer(const KmTimer&) = delete;
  KmTimer& operator=(const KmTimer&) = delete;
};
void km_timers_collect();  // call after a stream synchronisation: folds every finished stretch into its accumulator

// counts[k], starts[k+1] (exclusive prefix), sorted_ids[n] grouped by cluster, ascending inside.
int32_t km_group(const uint32_t* assign, uint32_t n, uint32_t k, uint32_t* sorted_ids, uint32_t* counts, uint32_t* starts,
                 KMeansScratch& ws, hipStream_t st);

// update_centroids (ivfflat.rs:47-71): Cnew[c] = (0 + x_i1 + x_i2 + ...) / count in ascending i, 0 if empty.
int32_t km_update(const float* X, uint32_t ldx, uint32_t d, const uint32_t* sorted_ids, const uint32_t* starts, const uint32_t* counts,
                  uint32_t k, float* Cnew, uint32_t ldc, hipStream_t st);
// The two halves of it for rows sharded over processes: S[c] += members of c among these rows, ascending (S carries the
// running sums of the ranks before this one); the last rank divides by the GLOBAL counts.
int32_t km_update_sums(const float* X, uint32_t ldx, uint32_t d, const uint32_t* sorted_ids, const uint32_t* starts, uint32_t k, float* S,
                       uint32_t ldc, hipStream_t st);
int32_t km_finish_centroids(const float* S, const uint32_t* counts, uint32_t k, uint32_t ldc, float* Cnew, hipStream_t st);

// calculate_kmeans_cost (ivfflat.rs:138-149): strict left-to-right f32 fold of mind[0..n) starting from *init_dev
// (nullptr = 0.0: the fold of a single process); result at *out_dev.
int32_t km_cost_fold(const float* mind, uint64_t n, const float* init_dev, float* out_dev, hipStream_t st);

// bitwise equality of two f32 arrays (to_hashkey comparison, ivfflat.rs:84-93); *flag_dev = 1 if ANY word differs.
int32_t km_differs(const float* a, const float* b, uint64_t n_words, uint32_t* flag_dev, hipStream_t st);

}  // namespace vers
