/*
 * vers_hip.h -- C ABI of libvers_hip.so: the MI355X (gfx950) IVFFlat hot path
 * that drops in behind ashrielbrian/vers's `Index` trait.
 *
 * Every entry point names the reference interface it replaces
 * (paths relative to /root/reference/vers/src).  The Rust-side binding a vers
 * maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - every function returns an int32 status (VERS_OK == 0); nothing unwinds
 *     across the boundary.  The reference panics where these return
 *     VERS_ERR_NAN / VERS_ERR_INSUFFICIENT / VERS_ERR_EMPTY; a Rust shim
 *     turns a non-zero status back into a panic to keep the behaviour;
 *   - vers_last_error() returns a thread-local message for the last failure;
 *   - pointers are HOST pointers unless the function name ends in `_dev`;
 *     `_dev` functions take device pointers plus a hipStream_t (as void*,
 *     NULL = the null stream), enqueue work and return without synchronising;
 *   - rows are f32, row-major; host row pitch is passed in BYTES so that a
 *     Rust Vec<Vector<N>> (#[repr(align(256))], base.rs:14-17, pitch =
 *     round_up(4*N, 256)) can be handed over without repacking;
 *   - ids are u64 (Rust usize), distances f32.  All distances are produced
 *     with the reference's own arithmetic (sequential f32 sum of (a-b)^2, no
 *     FMA), so they are bit-identical to the reference's, not merely close.
 */
#ifndef VERS_HIP_H
#define VERS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VERS_OK 0
#define VERS_ERR_INVALID 1      /* bad argument / unsupported size */
#define VERS_ERR_NAN 2          /* reference: partial_cmp().unwrap() panics on NaN */
#define VERS_ERR_INSUFFICIENT 3 /* reference: index out of bounds at ivfflat.rs:169 */
#define VERS_ERR_HIP 4          /* HIP runtime failure */
#define VERS_ERR_EMPTY 5        /* reference: min_by(..).unwrap() on zero centroids, ivfflat.rs:207 */
#define VERS_ERR_COMM 6         /* a vers_comm_t callback of the host reported failure */

#define VERS_METRIC_L2SQ 0    /* Vector::squared_euclidean, base.rs:119-126 (what IVFFlat uses) */
#define VERS_METRIC_COSDIST 1 /* Vector::cosine_similarity(normalized=true) = 1 - dot, base.rs:153-155 */

#define VERS_MAX_TOPK 64 /* one key per lane: the width of a result list inside the kernels.  NOT a limit of any entry point:
                          * every search takes any top_k (and nprobe) like the reference; beyond 64 the result comes 64 ranks
                          * per pass. */

/* Thread-local description of the most recent failure on this thread. */
const char* vers_last_error(void);
/* ABI version of this header (bumped on incompatible change). */
int32_t vers_abi_version(void);
/* Number of visible HIP devices. */
int32_t vers_device_count(int32_t* out_count);

/* ------------------------------------------------------------------------ *
 * Flat corpus: brute-force scan.                                            *
 * Replaces utils::search_exhaustive (utils.rs:68-82): every row scored with *
 * squared_euclidean, stable ascending sort (ties -> lower index), take k.   *
 * ------------------------------------------------------------------------ */
typedef struct vers_flat vers_flat_t;

int32_t vers_flat_create(int32_t device, uint32_t d, vers_flat_t** out);
int32_t vers_flat_destroy(vers_flat_t* h);
/* Copies n rows (pitch row_stride_bytes >= 4*d) into HBM; position = vec_id.  With vers_set_option("shadow", 1) (default) the
 * handle also keeps an fp16 shadow of the rows (+ 2 d + 8 bytes per row, optional memory: dropped when it does not fit): a single
 * query (b == 1, top_k <= 48) then streams the shadow -- half the bytes -- and is finished exactly (pre-selection, certificate,
 * exact re-score; exact re-scan when the certificate fails); vers_set_option("single_shadow", 0) keeps the f32 scan.  Same results. */
int32_t vers_flat_upload(vers_flat_t* h, const float* rows, uint64_t n, uint64_t row_stride_bytes);
/* Same from rows already in HBM (row-major, pitch ld_floats >= d).  The handle keeps its own
 * copy in the scan layout (lane-transposed 64-row tiles); the caller's buffer can be freed. */
int32_t vers_flat_upload_dev(vers_flat_t* h, const float* rows_dev, uint64_t n, uint64_t ld_floats);
/* b queries (pitch q_stride_bytes).  out_ids/out_dist: [b * top_k], row q at
 * q*top_k; out_count[q] = min(top_k, n) results written for query q.  Any top_k (utils.rs:79 `take(k)` has no cap). */
int32_t vers_flat_search(vers_flat_t* h, const float* queries, uint64_t q_stride_bytes, uint32_t b,
                         uint32_t top_k, uint32_t metric, uint64_t* out_ids, float* out_dist,
                         uint32_t* out_count);
/* Same, everything in HBM; queries pitch ldq_floats (any value >= d).  Errors
 * found on the device (NaN) are latched and reported by vers_flat_poll. */
int32_t vers_flat_search_dev(vers_flat_t* h, const float* queries_dev, uint64_t ldq_floats, uint32_t b,
                             uint32_t top_k, uint32_t metric, uint64_t* out_ids_dev, float* out_dist_dev,
                             uint32_t* out_count_dev, void* stream);
/* Synchronises `stream` and returns (then clears) the latched device status. */
int32_t vers_flat_poll(vers_flat_t* h, void* stream);
/* Duration in ms of the most recent scan-kernel launch of this handle
 * (HIP events on the launch stream; synchronises on those events). */
int32_t vers_flat_last_scan_ms(vers_flat_t* h, float* out_ms);

/* ------------------------------------------------------------------------ *
 * k-means primitives on host arrays (unit-test / integration surface; the   *
 * index build below chains the same kernels without leaving the device).    *
 * ------------------------------------------------------------------------ */
/* IVFFlatIndex::assign_to_clusters (ivfflat.rs:29-46): out_assign[i] = FIRST
 * argmin_c squared_euclidean(rows[i], centroids[c]).  out_min_dist (nullable)
 * receives that minimum.  k == 0 with n > 0 -> VERS_ERR_EMPTY; NaN distance
 * with k >= 2 -> VERS_ERR_NAN (the reference panics in both cases). */
int32_t vers_kmeans_assign(int32_t device, const float* rows, uint64_t n, uint64_t row_stride_bytes,
                           const float* centroids, uint64_t k, uint64_t c_stride_bytes, uint32_t d,
                           uint64_t* out_assign, float* out_min_dist);
/* The same on DEVICE-resident arrays (row-major, pitches in floats, multiples of 4, >= d; padding columns may hold
 * anything): out_assign_dev [n] u64 (the reference's usize), out_min_dist_dev [n] f32 or NULL.  Synchronous.  What a host
 * that streams a corpus larger than one GPU through a trained quantiser calls per chunk (cfg4: 100M rows in chunks).
 * The call's device scratch (centroid operands, workspaces: up to ~1 GiB at k = 65536) is kept per DEVICE between calls, under a
 * lock (calls on one device take turns); a call with n == 0 releases it. */
int32_t vers_kmeans_assign_dev(int32_t device, const float* rows_dev, uint64_t n, uint64_t ld_floats,
                               const float* centroids_dev, uint64_t k, uint64_t c_ld_floats, uint32_t d,
                               uint64_t* out_assign_dev, float* out_min_dist_dev);
/* Diagnostics of the matrix-core assign path (large builds; VERS_ASSIGN=1 forces the exact scan, =2 the
 * matrix cores): process-wide number of points assigned through it and how many of those failed the
 * certificate and were re-done by the exact scan.  Results are bit-identical either way. */
int32_t vers_assign_stats(uint64_t* out_points, uint64_t* out_fallbacks, int32_t reset);
/* Measurement hook: where the time of this process's build_index / k-means calls went (HIP events on the build's
 * stream).  out[8]: [0] ms in the assign contraction launches (queries x centroids on the matrix cores, ivfflat.rs:29-46),
 * [1] their count, [2] their algorithmic flop (2 * points * k * d), [3] wall ms of whole matrix-core assign passes (contraction
 * + arg-min + exact re-score + exact re-scans), [4] passes, [5] ms in update_centroids (:47-71), [6] ms in the cost fold
 * (:138-149), [7] points whose certificate failed and were settled by the exact kernels.  reset != 0 zeroes them. */
int32_t vers_build_stats(double* out8, int32_t reset);
/* The same builds by PHASE, host wall clock in ms (the build synchronises at every phase boundary anyway).  out[10]:
 * [0] whole build_index calls, [1] the build's device allocations, [2] assign passes (all of them; [3] the FIRST since the
 * last reset -- it pays the process's cold start: code load, first launches, first touch of 10s of GB -- and [4] their count),
 * [5] update_centroids, [6] the cost fold, [7] the inverted lists (grouping, storage plan + allocation, row placement; sharded:
 * + the rows-to-owners exchange), [8] what derives from the stored rows (centroid operands, |x|^2, fp16 shadow, row-major
 * copy), [9] the rest (init draws, convergence tests, read-back).  reset != 0 zeroes them (and vers_build_stats'). */
int32_t vers_build_phases(double* out10, int32_t reset);
/* IVFFlatIndex::update_centroids (ivfflat.rs:47-71): per cluster the f32 sum
 * of its members in ascending row order divided by the count; empty cluster
 * -> zero vector.  out_centroids is packed [k * d]. */
int32_t vers_kmeans_update(int32_t device, const float* rows, uint64_t n, uint64_t row_stride_bytes,
                           const uint64_t* assign, uint64_t k, uint32_t d, float* out_centroids);
/* IVFFlatIndex::calculate_kmeans_cost (ivfflat.rs:138-149): left-to-right f32
 * fold of squared_euclidean(rows[i], centroids[assign[i]]). */
int32_t vers_kmeans_cost(int32_t device, const float* rows, uint64_t n, uint64_t row_stride_bytes,
                         const float* centroids, uint64_t k, uint64_t c_stride_bytes, const uint64_t* assign,
                         uint32_t d, float* out_cost);

/* ------------------------------------------------------------------------ *
 * IVFFlat index.  The handle is the DEVICE CACHE of the five fields of       *
 * IVFFlatIndex<N> (ivfflat.rs:8-15); the host side (Rust) keeps owning        *
 * values / centroids / assignments / ids for serde and rebuilds the cache    *
 * with vers_ivf_upload after Index::load_index (base.rs:45-58).              *
 * ------------------------------------------------------------------------ */
typedef struct vers_ivf vers_ivf_t;

int32_t vers_ivf_create(int32_t device, uint32_t d, vers_ivf_t** out);
int32_t vers_ivf_destroy(vers_ivf_t* h);
/* Metric of the index (extension; set before build / upload).  VERS_METRIC_L2SQ (default) is what
 * the reference's IVFFlat computes everywhere (ivfflat.rs:37-38,146,159,175,205).  VERS_METRIC_COSDIST puts
 * Vector::cosine_similarity(normalized=true) = 1 - dot_product (base.rs:91-93,153-155: sequential f32 dot) in each of
 * those places -- assign_to_clusters, the k-means cost, add, the ranking of the lists and the scoring of their rows
 * -- with the same first-minimum / stable-sort rules; centroids stay plain means (not re-normalised), as in the
 * reference.  Results carry the bits of the sequential f32 arithmetic in both metrics. */
int32_t vers_ivf_set_metric(vers_ivf_t* h, uint32_t metric);
int32_t vers_ivf_get_metric(vers_ivf_t* h, uint32_t* out_metric);

/* IVFFlatIndex::build_index(num_clusters, num_attempts, max_iterations, &vectors)
 * (ivfflat.rs:102-136), k-means included (build_kmeans :73-100, cost :138-149).
 * init_indices [num_attempts * num_clusters]: the draws of initialize_centroids
 * (ivfflat.rs:18-27: k indices WITH replacement from an unseeded thread_rng) are made
 * by the caller, so the Rust shim keeps using rand::thread_rng and tests can inject.
 * Outputs (host, caller-owned, nullable): out_centroids k rows of pitch c_stride_bytes
 * (>= 4*d; a Vec<Vector<N>> is written in place with its own pitch round_up(4*N, 256),
 * bytes between rows are left untouched), out_assignments [n], out_cost (best cost),
 * out_kept (0 when no attempt beat +inf, e.g. num_attempts == 0: the index then has
 * EMPTY centroids/assignments exactly like the reference), out_iterations
 * [num_attempts] loop bodies executed per attempt. */
int32_t vers_ivf_build(vers_ivf_t* h, const float* rows, uint64_t n, uint64_t row_stride_bytes,
                       uint64_t num_clusters, uint64_t num_attempts, uint64_t max_iterations,
                       const uint64_t* init_indices, float* out_centroids, uint64_t c_stride_bytes,
                       uint64_t* out_assignments, float* out_cost, int32_t* out_kept, uint64_t* out_iterations);
/* Same with the vectors already in HBM (row-major, pitch ld_floats >= d, a multiple of 4; columns
 * d .. ld_floats-1 are the caller's padding: never read into the index, they may hold anything). */
int32_t vers_ivf_build_dev(vers_ivf_t* h, const float* rows_dev, uint64_t n, uint64_t ld_floats,
                           uint64_t num_clusters, uint64_t num_attempts, uint64_t max_iterations,
                           const uint64_t* init_indices, float* out_centroids, uint64_t c_stride_bytes,
                           uint64_t* out_assignments, float* out_cost, int32_t* out_kept, uint64_t* out_iterations);
/* Rebuilds the device cache from the host fields (after Index::load_index): values,
 * centroids, assignments; ids[c] is implied (ascending positions with assignments == c,
 * which is what build_index + add produce). */
int32_t vers_ivf_upload(vers_ivf_t* h, const float* rows, uint64_t n, uint64_t row_stride_bytes,
                        const float* centroids, uint64_t k, uint64_t c_stride_bytes,
                        const uint64_t* assignments);
/* The same from DEVICE-resident fields (a host that keeps `values` in HBM, or re-shards one built index over another
 * world with vers_ivf_set_shard): rows row-major with pitch ld_floats >= d (a multiple of 4; the padding columns may hold
 * anything), centroids with pitch c_ld_floats >= d, assignments as the reference's usize = u64.  Synchronous. */
int32_t vers_ivf_upload_dev(vers_ivf_t* h, const float* rows_dev, uint64_t n, uint64_t ld_floats,
                            const float* centroids_dev, uint64_t k, uint64_t c_ld_floats,
                            const uint64_t* assignments_dev);
/* STREAMED rebuild of the device cache: the reference's save -> load -> search sequence (utils.rs:140-148, base.rs:45-58)
 * for an index that no single GPU can hold.  The five fields (ivfflat.rs:8-15) arrive in chunks; a handle sharded with
 * vers_ivf_set_shard keeps only the rows of the lists it owns, with their GLOBAL vec ids, and no buffer of n_total rows
 * exists anywhere (device memory = the owned lists + one bounded staging chunk).  vers_ivf_upload is this sequence.
 *   begin : centroids [k] (pitch c_stride_bytes), list_lengths [k] = ids[c].len() of every list (ALL lists, also the ones
 *           another rank owns: the search plans with global lengths), n_total = assignments.len().  The handle holds no
 *           index until _end succeeds (searches in between see an empty index).
 *   chunk : rows of vec ids first_vec_id .. first_vec_id + n - 1 and their assignments.  Chunks arrive in ascending,
 *           contiguous order (first_vec_id == rows seen so far; checked): a list then fills in ascending vec id, which is
 *           what build_index + add produce (ivfflat.rs:123-127, 209-211).  Any chunk size; host rows are staged through a
 *           bounded pinned buffer, and only the rows this rank owns cross PCIe.
 *   end   : checks that n_total rows arrived and that every list received exactly list_lengths[c] of them
 *           (VERS_ERR_INVALID otherwise, the handle stays empty), then derives |x|^2, the shadow and the row-major copy.
 * Any other build / upload / add on the handle abandons a streamed upload in progress. */
int32_t vers_ivf_upload_begin(vers_ivf_t* h, const float* centroids, uint64_t k, uint64_t c_stride_bytes,
                              const uint64_t* list_lengths, uint64_t n_total);
int32_t vers_ivf_upload_chunk(vers_ivf_t* h, const float* rows, uint64_t row_stride_bytes, const uint64_t* assignments,
                              uint64_t first_vec_id, uint64_t n);
/* rows_dev row-major with pitch ld_floats >= d (a multiple of 4; padding columns may hold anything), assignments_dev u64. */
int32_t vers_ivf_upload_chunk_dev(vers_ivf_t* h, const float* rows_dev, uint64_t ld_floats, const uint64_t* assignments_dev,
                                  uint64_t first_vec_id, uint64_t n);
int32_t vers_ivf_upload_end(vers_ivf_t* h);
/* Index::add (ivfflat.rs:200-213): nearest centroid by first minimum; the new vector gets
 * vec_id = assignments.len() (the reference ignores the caller's vec_id, :209) and is appended
 * to that list.  Returns both so the host can mirror values/assignments/ids. */
int32_t vers_ivf_add(vers_ivf_t* h, const float* row, uint64_t* out_cluster, uint64_t* out_vec_id);
/* Index::search_approximate (ivfflat.rs:153-198) for b queries.
 *   nprobe == 0 : the reference's semantics -- nearest list, spill into the next-nearest while
 *                 fewer than top_k results; results CONCATENATED per list (not globally sorted);
 *                 fewer than top_k vectors reachable -> VERS_ERR_INSUFFICIENT (reference panics).
 *   nprobe >= 1 : extension named by BASELINE.json (not in the reference): all rows of the nprobe
 *                 nearest lists, one global stable order by (distance, probe rank, list position).
 * out_ids/out_dist [b*top_k], out_count[q] results for query q.  Any top_k and nprobe (like the reference: its walk has
 * no cap); top_k <= 54 and nprobe <= 64 is the fast domain (matrix-core list scan), wider results take one ordered-chain
 * pass per 64 ranks.  Reference mode follows the spill through ALL lists if it must (like the reference's walk): the
 * host-pointer call retries deeper, a device-pointer call -- which cannot come back -- ranks up front as many lists as the
 * list lengths can make the walk need (48 unless the index holds that many near-empty lists). */
int32_t vers_ivf_search(vers_ivf_t* h, const float* queries, uint64_t q_stride_bytes, uint32_t b,
                        uint32_t top_k, uint32_t nprobe, uint64_t* out_ids, float* out_dist,
                        uint32_t* out_count);
int32_t vers_ivf_search_dev(vers_ivf_t* h, const float* queries_dev, uint64_t ldq_floats, uint32_t b,
                            uint32_t top_k, uint32_t nprobe, uint64_t* out_ids_dev, float* out_dist_dev,
                            uint32_t* out_count_dev, void* stream);
/* Synchronises `stream`, returns and clears the status latched by _dev calls. */
int32_t vers_ivf_poll(vers_ivf_t* h, void* stream);
/* utils::search_exhaustive (utils.rs:68-82) over the index's own values (ties -> lower vec_id);
 * the recall ground truth. */
int32_t vers_ivf_search_exhaustive(vers_ivf_t* h, const float* queries, uint64_t q_stride_bytes, uint32_t b,
                                   uint32_t top_k, uint32_t metric, uint64_t* out_ids, float* out_dist,
                                   uint32_t* out_count);
int32_t vers_ivf_search_exhaustive_dev(vers_ivf_t* h, const float* queries_dev, uint64_t ldq_floats,
                                       uint32_t b, uint32_t top_k, uint32_t metric, uint64_t* out_ids_dev,
                                       float* out_dist_dev, uint32_t* out_count_dev, void* stream);
/* n = vectors in the index (assignments.len()), k = centroids, longest list. */
int32_t vers_ivf_info(vers_ivf_t* h, uint64_t* out_n, uint64_t* out_k, uint64_t* out_max_list_len);
int32_t vers_ivf_list_lengths(vers_ivf_t* h, uint64_t* out_lengths /* [k] */);
/* Measurement hook: the most recent inverted-list scan launch -- duration (HIP events on its
 * stream), rows of the union of probed lists (algorithmic), rows actually streamed, work items. */
int32_t vers_ivf_last_scan(vers_ivf_t* h, float* out_ms, uint64_t* out_union_rows,
                           uint64_t* out_streamed_rows, uint32_t* out_items);

/* Throughput serving loops (extension; no reference counterpart -- ivfflat.rs:155-161 ranks the lists inside the
 * search itself): note the queries of the NEXT batch; the following search on this handle (the current batch) then
 * stages them and ranks their lists on a side stream of the handle right behind its own list-scan launch, i.e. under its
 * exact finish, which leaves the chip mostly idle.  A later vers_ivf_search_dev / vers_ivf_search_partial_dev call with
 * the SAME queries_dev, ldq_floats, b and nprobe picks the prepared result up (same bits) instead of computing it; the
 * query block must not change in between.  Two batches can be prepared at a time; a no-op for nprobe == 0, for more than
 * 48 probes and for batches the coarse quantiser does not run on the matrix cores for (b < 32).  `stream` is ignored
 * (kept for ABI stability): the work is ordered on the stream of the search that starts it. */
int32_t vers_ivf_coarse_ahead_dev(vers_ivf_t* h, const float* queries_dev, uint64_t ldq_floats, uint32_t b, uint32_t nprobe, void* stream);

/* ---- one process per GPU: the corpus shards BY CLUSTER ------------------------------------------
 * Every rank holds the centroids and all list LENGTHS, but stores only the lists it owns.  A search
 * runs the (cheap, replicated) coarse quantiser everywhere, scans the local lists, and yields a
 * partial top-k as (key, vec_id) pairs; key = (order-preserving distance bits << 32) | sequence
 * number of the row in the query's probe order, identical on every rank, so that ONE all-gather of
 * [b][top_k] pairs (RCCL, done by the caller) followed by vers_topk_merge_dev reproduces the
 * single-GPU result bit for bit.  No reference analogue (the reference is single-process). */
/* Deterministic LPT plan (host only, no GPU): owner rank of every list from the list lengths. */
int32_t vers_shard_plan(const uint64_t* list_lengths, uint64_t k, uint32_t world, uint8_t* out_owner);
/* Before build/upload: this handle keeps only the lists vers_shard_plan gives to `rank`. */
int32_t vers_ivf_set_shard(vers_ivf_t* h, uint32_t rank, uint32_t world);
int32_t vers_ivf_owners(vers_ivf_t* h, uint8_t* out_owner /* [k] */);
/* ---- build_index over a ROW-SHARDED corpus (one process per GPU; no process ever holds all rows) ----------
 * The library owns no communicator: the host hands it one as a table of callbacks over device buffers (RCCL over
 * xGMI in vers_amd/dist.py through torch.distributed; a Rust host would put its own RCCL communicator behind the
 * same five functions).  Every callback is SYNCHRONOUS from the library's point of view: the library has
 * synchronised its stream before the call, and the data is in place when the callback returns 0.
 *   all_gather   : recv_dev[r*bytes .. (r+1)*bytes) = rank r's send_dev[0 .. bytes)
 *   send / recv  : point to point with `peer` (the chain of running sums below)
 *   broadcast    : buf_dev of `root` to everyone
 *   all_to_all_v : byte counts / offsets per peer, [world] each (rows to the owners of their lists) */
typedef struct vers_comm {
  void* ctx;
  uint32_t rank, world;
  int32_t (*all_gather)(void* ctx, const void* send_dev, void* recv_dev, uint64_t bytes);
  int32_t (*send)(void* ctx, const void* buf_dev, uint64_t bytes, uint32_t peer);
  int32_t (*recv)(void* ctx, void* buf_dev, uint64_t bytes, uint32_t peer);
  int32_t (*broadcast)(void* ctx, void* buf_dev, uint64_t bytes, uint32_t root);
  int32_t (*all_to_all_v)(void* ctx, const void* send_dev, const uint64_t* send_bytes, const uint64_t* send_off,
                          void* recv_dev, const uint64_t* recv_bytes, const uint64_t* recv_off);
} vers_comm_t;
/* IVFFlatIndex::build_index (ivfflat.rs:102-136) where rank r holds rows [row_begin, row_begin + n_local) of the
 * n_total vectors (contiguous ascending ranges in rank order; checked).  Bit-identical to the single-process build:
 *   assign_to_clusters (:29-46)  every rank assigns its own rows -- no exchange;
 *   update_centroids   (:47-71)  the reference adds the members of a cluster in ascending vec_id; rank r CONTINUES
 *                                rank r-1's per-cluster running sums over its own range (k*d*4 bytes hop to the next
 *                                rank, 12.6 MB at k=4096 d=768), the last rank divides by the all-gathered counts
 *                                and broadcasts the centroids.  (A reduce-scatter of partial sums re-associates the
 *                                f32 additions and cannot match the reference's order.)
 *   cost               (:138-149) the f32 fold over the points is chained through the ranks the same way (4 bytes);
 *   lists              (:123-127) lists are dealt to ranks by vers_shard_plan over the global lengths; every rank
 *                                sends each row to the owner of its list (ONE all_to_all_v), where rows arrive in
 *                                rank order = ascending vec_id.
 * init_indices are GLOBAL row indices.  On return the handle is sharded by cluster exactly like
 * vers_ivf_set_shard(rank, world) + build: centroids and all list lengths everywhere, rows of the owned lists only.
 * out_assignments_local (host, nullable): [n_local] clusters of this rank's rows. */
int32_t vers_ivf_build_sharded_dev(vers_ivf_t* h, const float* rows_dev, uint64_t n_local, uint64_t ld_floats,
                                   uint64_t row_begin, uint64_t n_total, uint64_t num_clusters, uint64_t num_attempts,
                                   uint64_t max_iterations, const uint64_t* init_indices, const vers_comm_t* comm,
                                   uint64_t* out_assignments_local, float* out_cost, int32_t* out_kept,
                                   uint64_t* out_iterations);
/* Process-wide switches: every one is a named option set here (or, for a process one does not control from inside, through the ONE
 * environment variable VERS_OPTIONS="name=value,name=value", read once; besides it the library reads only VERS_SHADOW and
 * VERS_ROWMAJOR, the two memory switches of INTEGRATION.md = options "shadow" / "rowmajor").  Unknown name: VERS_ERR_INVALID.
 * Results are bit-identical under every setting: the switches choose between exact paths and pre-filters behind exact finishes.
 *  Memory:
 *   "shadow" (1)        indexes built / uploaded from now on keep an fp16 shadow of their rows for the list scans
 *                       (vers_ivf_shadow_state); 0 = none, and searches on existing handles read their f32 rows until it is 1 again.
 *   "rowmajor" (-1)     the row-major f32 copy the exact finish gathers from: -1 = kept while the rows take <= 1/4 of the device
 *                       (and "memory" is 0), 0 never, 1 always.
 *   "memory" (0)        0 = speed: f32 tiles + fp16 shadow + row-major f32 copy (2.7 x the f32 rows at d = 768); 1 = compact: ONE f32
 *                       copy -- tiles + shadow, 1.7 x --, the exact finish gathers its survivors from the tiles (16-byte pieces).
 *                       Read when an index is built / uploaded.
 *  Serving:
 *   "single_shadow" (1) a single query (b == 1, nprobe >= 1) streams the fp16 shadow and is finished exactly like a batch;
 *                       0 = the ordered-chain scan of the f32 rows.
 *   "host_spin" (1)     a host-pointer single-query call waits by spinning on the pinned status word (<= 2 ms); 0 = hipStreamSynchronize.
 *   "pre_min_batch" (4) the smallest batch whose list scan runs on the matrix cores when its lists are shared by fewer than two
 *                       queries on average; smaller batches run as consecutive single queries.
 *   "scan_reserve_cus" (-1 = auto: 64 while another batch of the handle is in flight on another stream, else 0) compute units the
 *                       persistent matrix-core list scan leaves free for the other batches' latency-bound kernels and the exchange.
 *   "scan_events" (2)   HIP event records around list-scan launches (vers_ivf_last_scan / vers_ivf_scan_times): 1 always, 0 never,
 *                       2 for batches only (the two records cost a single-query call 5.5-6 us).
 *  A/B and forced paths (tests, measurements):
 *   "gemm_x3" (3)       bit 0 the k-means assign contraction, bit 1 the coarse quantiser's contraction as three bf16 MFMA products of
 *                       hi/lo-split operands instead of the f32 MFMA kernel.
 *   "prescan" (1)       0 = ordered chains for batches too, 2 = every list-scan certificate fails (the exact re-scan runs).
 *   "coarse" (0)        1 = the batched coarse quantiser always exact, 2 = every coarse certificate fails.
 *   "assign" (0)        k-means assign: 1 = never on the matrix cores, 2 = always (default: from 1e11 flop per pass).
 *   "assign_terms" (0)  the assign contraction's first filter: 1 = ONE product of fp16 operands (a third of the MFMAs; the points it leaves
 *                       open go to the tile-limited exact re-scan), 3 = all three products of the bf16 hi/lo split, 0 = probe per build.
 *   "assign_glds" (-1)  the one-product filter with both operands fp16 in memory, staged by LDS-DMA in whole cache lines: 1 = always, 0 = never
 *                       (the register-staged kernel that converts the f32 batch while it stages it), -1 = from 4096 centroids on.
 *   "pre_narrow" (0), "pre_wide" (1), "pre_hi_only" (0)   the list scan's query-block width (16 / 64 queries) and fp16 hi-only query blocks.
 *   "wide_k" (1)        results of 49 .. 200 keys (batches, nprobe >= 1) stay on the matrix-core list scan with candidate lists four keys
 *                       per lane wide; 0 = the ordered chains, 64 ranks per pass.
 *   "coarse1" (1), "scan1t" (1), "ref_as_nprobe1" (1), "assign_tiles" (1), "assign_tiles_min" (64), "seg_rows" (0), "pre_slack" (0),
 *   "upload_stage_mb" (256)   kernel-choice and sizing knobs of DESIGN.md section 5.
 *   "scan_debug" (0), "poison_alloc" (-1), "poison_slack_bits" (-1), "test_fail_sharded" (0)   diagnosis: phase stamps / skipped
 *                       phases, new device buffers filled with a byte, slack rows filled with an f32 bit pattern, the next n sharded
 *                       searches fail locally. */
int32_t vers_set_option(const char* name, int64_t value);
/* Device memory the library holds in this process right now (rows, ids, scratch of every handle) and its high-water
 * mark since the last reset -- lets a test assert that no rank of a sharded build ever allocated the whole corpus. */
int32_t vers_mem_stats(uint64_t* out_bytes_now, uint64_t* out_bytes_peak, int32_t reset_peak);
/* Local part of search_approximate: out_keys/out_ids [b*top_k], kKeyMax (all ones) padded. */
int32_t vers_ivf_search_partial_dev(vers_ivf_t* h, const float* queries_dev, uint64_t ldq_floats, uint32_t b,
                                    uint32_t top_k, uint32_t nprobe, uint64_t* out_keys_dev,
                                    uint64_t* out_ids_dev, void* stream);
/* Local part of utils::search_exhaustive (utils.rs:68-82) over the rows this rank stores: keys = (order-preserving
 * distance bits << 32) | vec_id -- the reference's stable order, ties to the lower index -- so the same all-gather +
 * vers_topk_merge_dev (any nprobe >= 1: global top-k by key) yields the brute-force result of the whole corpus. */
int32_t vers_ivf_search_exhaustive_partial_dev(vers_ivf_t* h, const float* queries_dev, uint64_t ldq_floats, uint32_t b,
                                               uint32_t top_k, uint32_t metric, uint64_t* out_keys_dev,
                                               uint64_t* out_ids_dev, void* stream);
/* Merge of the gathered partials into final results: rank r's keys at keys_dev + r*rank_stride
 * ([b][top_k], rank_stride in elements), ids likewise -- so one all-gathered [world][2][b][top_k]
 * buffer serves both with rank_stride = 2*b*top_k. */
int32_t vers_topk_merge_dev(const uint64_t* keys_dev, const uint64_t* ids_dev, uint64_t rank_stride, uint32_t world,
                            uint32_t b, uint32_t top_k, uint32_t nprobe, uint64_t* out_ids_dev,
                            float* out_dist_dev, uint32_t* out_count_dev, void* stream);

/* Bytes of the three copies of the stored rows this handle holds: the f32 tiles (always), the fp16 shadow the batched
 * list scan streams (vers_ivf_shadow_state), the row-major f32 copy the exact finish gathers from (kept whenever the rows
 * take at most a quarter of the device's memory; VERS_ROWMAJOR=0 switches it off).  0 = that copy does not exist. */
int32_t vers_ivf_layout_bytes(vers_ivf_t* h, uint64_t* out_rows, uint64_t* out_shadow, uint64_t* out_rowmajor);

/* ---- sharded search WITHOUT the host in the loop ----------------------------------------------------------------
 * The ONE exchange of a sharded search (all-gather of the per-rank partial top-k) as a stream-ordered
 * callback: all_gather_async queues, on `stream`, an all-gather of `bytes` bytes per rank from send_dev into recv_dev
 * ([world][bytes], rank order) and returns without synchronising.  include/vers_comm_rccl.h fills one from an
 * ncclComm_t (ncclAllGather on the batch's stream: RCCL over xGMI); tests put gloo behind the same signature. */
typedef struct vers_gather {
  void* ctx;
  uint32_t rank, world;
  int32_t (*all_gather_async)(void* ctx, const void* send_dev, void* recv_dev, uint64_t bytes, void* stream);
} vers_gather_t;
/* search_approximate over lists sharded by cluster (vers_ivf_set_shard / vers_ivf_build_sharded_dev), end to end on
 * `stream`: local partial search -> g->all_gather_async of [2][b][top_k] u64 (keys | ids) -> merge of the world's
 * partials (vers_topk_merge_dev) into out_*_dev.  Nothing synchronises; the gather buffers belong to the workspace the
 * call leases (one per stream in flight), so batches kept in flight on several streams overlap.  Every rank must issue
 * its calls in the same order.  g == NULL: a single process, no exchange (same as vers_ivf_search_dev).  Results are bit-identical to the
 * unsharded index; statuses latch per stream (vers_ivf_poll). */
int32_t vers_ivf_search_sharded_dev(vers_ivf_t* h, const vers_gather_t* g, const float* queries_dev, uint64_t ldq_floats,
                                    uint32_t b, uint32_t top_k, uint32_t nprobe, uint64_t* out_ids_dev,
                                    float* out_dist_dev, uint32_t* out_count_dev, void* stream);
/* utils::search_exhaustive (utils.rs:68-82) over the rows sharded across the ranks, the same way. */
int32_t vers_ivf_search_exhaustive_sharded_dev(vers_ivf_t* h, const vers_gather_t* g, const float* queries_dev,
                                               uint64_t ldq_floats, uint32_t b, uint32_t top_k, uint32_t metric,
                                               uint64_t* out_ids_dev, float* out_dist_dev, uint32_t* out_count_dev,
                                               void* stream);

/* Measurement hook: durations (HIP events on the search's stream) of the two kernels of the most recent batched
 * coarse quantiser -- the queries x centroids contraction on the f32 matrix cores (2*b*k*d flop) and the selection /
 * exact re-score / certificate kernel behind it. */
int32_t vers_ivf_last_coarse_ms(vers_ivf_t* h, float* out_gemm_ms, float* out_select_ms);
/* Measurement hook: duration (HIP events on the search's stream) of the exact finish -- ivf_rescore_kernel: merge of the
 * partial candidate lists, certificate, exact re-score of the survivors, emit -- of the most recent matrix-core batch. */
int32_t vers_ivf_last_finish_ms(vers_ivf_t* h, float* out_ms);
/* Batched coarse quantiser statistics: batches that went through the MFMA pre-selection (f32 matrix cores +
 * exact re-score + certificate, csrc/gemm.hip.h) and queries whose certificate failed and were re-done exactly. */
int32_t vers_ivf_coarse_stats(vers_ivf_t* h, uint64_t* out_mfma_batches, uint64_t* out_fallback_queries);
/* Batched list scan statistics (nprobe mode): batches whose list scan ran on the matrix cores (pre-selection +
 * exact re-score + certificate, csrc/prescan.hip.h) and queries whose certificate failed and were re-scanned
 * exactly.  Results are bit-identical either way; option "prescan" = 0 keeps the ordered-chain scan for every batch. */
int32_t vers_ivf_prescan_stats(vers_ivf_t* h, uint64_t* out_batches, uint64_t* out_fallback_queries);
/* An fp16 shadow copy of the stored rows (+50 % corpus memory) feeds the matrix-core pre-selection of batched searches:
 * half the HBM bytes per list scan, a ~2x wider certificate window, the same exact f32 finish -- results stay
 * bit-identical.  On by default (VERS_SHADOW=0 or vers_set_option("shadow", 0) before build / upload: f32 rows feed the
 * scan); skipped when its allocation fails; switched off for the handle when more than 1/8 of the queries failed the
 * certificate (data with many near-ties, or elements outside fp16's range).  out_active: 1 = in use. */
int32_t vers_ivf_shadow_state(vers_ivf_t* h, int32_t* out_active, uint64_t* out_bytes);
/* Durations (ms) of the most recent list-scan launches, oldest first (ring of 64); reset != 0
 * empties the ring.  Lets bench.py time every launch of the timed region without stalling it. */
int32_t vers_ivf_scan_times(vers_ivf_t* h, float* out_ms, uint32_t cap, uint32_t* out_n, int32_t reset);
/* Reads inverted list `cluster` back to the host in list order: rows (pitch row_stride_bytes) and
 * vec ids; out_rows / out_ids may be NULL to query the length only.  (ids[c] and values of the
 * reference, for host-side serde and for the CPU baseline of bench.py.) */
int32_t vers_ivf_get_list(vers_ivf_t* h, uint64_t cluster, float* out_rows, uint64_t row_stride_bytes,
                          uint64_t* out_ids, uint64_t cap_rows, uint64_t* out_len);
int32_t vers_ivf_get_centroids(vers_ivf_t* h, float* out_centroids, uint64_t c_stride_bytes);

/* ------------------------------------------------------------------------ *
 * Measurement tooling (bench.py): deterministic synthetic vectors written   *
 * straight into HBM; bit-identical to tests/datagen.py on the host.          *
 * kind 0 = uniform on the sphere, kind 1 = clustered (n_modes centres drawn  *
 * as kind 0 from seed_centres, row = normalize(centre[row % n_modes] +       *
 * sigma * noise)).  Rows start_row .. start_row+n-1 of the stream `seed`.    *
 * ------------------------------------------------------------------------ */
int32_t vers_gen_rows_dev(float* out_dev, uint64_t n, uint32_t d, uint64_t ld_floats, uint32_t kind,
                          uint64_t seed, uint64_t seed_centres, uint32_t n_modes, float sigma,
                          uint64_t start_row, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VERS_HIP_H */
