/*
 * vers_oracle.c -- CPU restatement of the IVFFlat hot path of ashrielbrian/vers.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under vers_amd/ may link, import or call
 * this file.  It exists so that tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py have something to check / time the HIP path
 * against.
 *
 * PARITY STATUS: "parity unpinned" by the reference's own tests -- the
 * reference has no tests, golden vectors or fixtures for this path (SURVEY.md
 * section 4), and the Rust crate cannot be built here (no rustc/cargo, needs
 * nightly + un-vendored crates).  The oracle is pinned instead by (a) hand
 * computable micro-cases in tests/test_oracle_micro.py and (b) bit-for-bit
 * agreement with an independent NumPy-float32 restatement (oracle/np_oracle.py)
 * on the committed fixtures under tests/golden/.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/vers/src).  Arithmetic rules that make it order-faithful:
 * IEEE f32, round-to-nearest, NO fma contraction, NO re-association; build
 * with  gcc -O2 -ffp-contract=off -fno-fast-math  (see oracle/Makefile).
 *
 * Error convention: the reference panics (partial_cmp().unwrap() on NaN,
 * slice index out of bounds, unwrap on None).  Here a "panic" is a negative
 * return code:
 *   VO_ERR_NAN          -2   a NaN distance reached a comparison
 *   VO_ERR_INSUFFICIENT -3   search ran out of clusters before top_k results
 *   VO_ERR_EMPTY        -4   min_by over zero centroids (.unwrap() on None)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define VO_OK 0
#define VO_ERR_NAN (-2)
#define VO_ERR_INSUFFICIENT (-3)
#define VO_ERR_EMPTY (-4)

/* ---- Vector<N> arithmetic: indexes/base.rs ------------------------------ */

/* base.rs:119-126  squared_euclidean: map (a-b).powi(2) then .sum()  --
 * a strictly sequential f32 fold, products rounded before the add. */
float vo_squared_euclidean(const float *a, const float *b, size_t d) {
  float acc = 0.0f;
  for (size_t i = 0; i < d; ++i) {
    float t = a[i] - b[i];
    float t2 = t * t;
    acc = acc + t2;
  }
  return acc;
}

/* base.rs:91-93  dot_product: map a*b then sum::<f32>() */
float vo_dot(const float *a, const float *b, size_t d) {
  float acc = 0.0f;
  for (size_t i = 0; i < d; ++i) {
    float p = a[i] * b[i];
    acc = acc + p;
  }
  return acc;
}

/* base.rs:95-97  magnitude = sqrt(dot(a,a)) */
float vo_magnitude(const float *a, size_t d) { return sqrtf(vo_dot(a, a, d)); }

/* base.rs:99-105 normalize (+ divide_by_scalar base.rs:74-83): magnitude<1e-6
 * returns the input unchanged; otherwise true division element by element. */
void vo_normalize(const float *a, float *out, size_t d) {
  float m = vo_magnitude(a, d);
  if (m < 1e-6f) {
    memmove(out, a, d * sizeof(float));
    return;
  }
  for (size_t i = 0; i < d; ++i) out[i] = a[i] / m;
}

/* base.rs:153-155  cosine_similarity(normalized=true) = 1 - dot */
float vo_cosine_distance(const float *a, const float *b, size_t d) {
  return 1.0f - vo_dot(a, b, d);
}

/* The IVF functions below take `metric`: 0 = squared_euclidean (what ivfflat.rs calls everywhere: :37-38,146,159,
 * 175,205), 1 = cosine distance 1 - dot (base.rs:153-155) in the same places -- the "cosine/dot" IVF of SURVEY.md
 * 8f-3, an EXTENSION: the reference's IVFFlat has no metric switch.  Everything else (first-minimum, stable sorts,
 * spill walk, ascending-order sums, no re-normalisation of centroids) is the reference's code path unchanged. */
static float vo_dist(const float *a, const float *b, size_t d, int metric) {
  return metric == 0 ? vo_squared_euclidean(a, b, d) : vo_cosine_distance(a, b, d);
}

/* ---- stable sort of (index, distance) pairs ------------------------------
 * itertools::sorted_by == collect + slice::sort_by (stable).  Any stable sort
 * yields the same permutation under a total order; NaN makes
 * partial_cmp().unwrap() panic as soon as it is compared, which (for n >= 2)
 * always happens, so NaN is detected up front. */
typedef struct {
  uint64_t id;
  float dist;
} vo_pair;

static void merge_sort_pairs(vo_pair *v, vo_pair *tmp, size_t n) {
  if (n < 2) return;
  size_t h = n / 2;
  merge_sort_pairs(v, tmp, h);
  merge_sort_pairs(v + h, tmp, n - h);
  size_t i = 0, j = h, o = 0;
  while (i < h && j < n) {
    /* take right only if strictly smaller: keeps equal keys in input order */
    if (v[j].dist < v[i].dist) tmp[o++] = v[j++];
    else tmp[o++] = v[i++];
  }
  while (i < h) tmp[o++] = v[i++];
  while (j < n) tmp[o++] = v[j++];
  memcpy(v, tmp, n * sizeof(vo_pair));
}

static int stable_sort_by_dist(vo_pair *v, size_t n) {
  if (n >= 2) {
    for (size_t i = 0; i < n; ++i)
      if (isnan(v[i].dist)) return VO_ERR_NAN;
  }
  vo_pair *tmp = (vo_pair *)malloc((n ? n : 1) * sizeof(vo_pair));
  merge_sort_pairs(v, tmp, n);
  free(tmp);
  return VO_OK;
}

/* ---- utils.rs:68-82  search_exhaustive ---------------------------------- */
/* enumerate -> squared_euclidean(v, query) -> stable sort -> take(top_k).
 * metric 0 = squared L2 (the reference), 1 = cosine distance 1-dot
 * (base.rs:153-155; extension used for the "cosine/dot" wording of cfg2).
 * Returns the number of results (min(top_k, n)) or a negative error. */
int64_t vo_search_exhaustive(const float *data, uint64_t n, uint64_t d, uint64_t ld,
                             const float *query, uint64_t top_k, int metric,
                             uint64_t *out_ids, float *out_dist) {
  vo_pair *p = (vo_pair *)malloc((n ? n : 1) * sizeof(vo_pair));
  for (uint64_t i = 0; i < n; ++i) {
    p[i].id = i;
    p[i].dist = metric == 0 ? vo_squared_euclidean(data + i * ld, query, d)
                            : vo_cosine_distance(data + i * ld, query, d);
  }
  int rc = stable_sort_by_dist(p, n);
  if (rc != VO_OK) {
    free(p);
    return rc;
  }
  uint64_t m = top_k < n ? top_k : n;
  for (uint64_t i = 0; i < m; ++i) {
    out_ids[i] = p[i].id;
    out_dist[i] = p[i].dist;
  }
  free(p);
  return (int64_t)m;
}

/* ---- ivfflat.rs:29-46  assign_to_clusters -------------------------------- */
/* min_by over centroids, comparator recomputes both distances; first minimum
 * wins (Iterator::min_by keeps the earlier element on Ordering::Equal);
 * NaN -> panic when k >= 2; k == 0 -> unwrap on None panics. */
int vo_assign_m(const float *X, uint64_t n, const float *C, uint64_t k, uint64_t d,
                uint64_t *out, int metric) {
  if (k == 0) return n ? VO_ERR_EMPTY : VO_OK;
  for (uint64_t i = 0; i < n; ++i) {
    const float *x = X + i * d;
    uint64_t best = 0;
    float bd = vo_dist(x, C, d, metric);
    if (k >= 2 && isnan(bd)) return VO_ERR_NAN;
    for (uint64_t c = 1; c < k; ++c) {
      float dc = vo_dist(x, C + c * d, d, metric);
      if (isnan(dc)) return VO_ERR_NAN;
      if (dc < bd) {
        bd = dc;
        best = c;
      }
    }
    out[i] = best;
  }
  return VO_OK;
}

int vo_assign(const float *X, uint64_t n, const float *C, uint64_t k, uint64_t d, uint64_t *out) {
  return vo_assign_m(X, n, C, k, d, out, 0);
}

/* ---- ivfflat.rs:47-71  update_centroids ---------------------------------- */
/* sums[c] = sums[c] + x in ascending data order (base.rs:62-72 add), counts;
 * centroid = sum / (count as f32) (base.rs:74-83); empty cluster -> zeros. */
void vo_update(const float *X, uint64_t n, const uint64_t *assign, uint64_t k, uint64_t d,
               float *Cout) {
  uint64_t *counts = (uint64_t *)calloc(k ? k : 1, sizeof(uint64_t));
  for (uint64_t i = 0; i < k * d; ++i) Cout[i] = 0.0f;
  for (uint64_t i = 0; i < n; ++i) {
    float *s = Cout + assign[i] * d;
    const float *x = X + i * d;
    for (uint64_t j = 0; j < d; ++j) s[j] = s[j] + x[j];
    counts[assign[i]] += 1;
  }
  for (uint64_t c = 0; c < k; ++c) {
    float *s = Cout + c * d;
    if (counts[c] > 0) {
      float cf = (float)counts[c];
      for (uint64_t j = 0; j < d; ++j) s[j] = s[j] / cf;
    } else {
      for (uint64_t j = 0; j < d; ++j) s[j] = 0.0f;
    }
  }
  free(counts);
}

/* ---- ivfflat.rs:138-149  calculate_kmeans_cost --------------------------- */
float vo_cost_m(const float *X, uint64_t n, const float *C, const uint64_t *assign, uint64_t d, int metric) {
  float acc = 0.0f;
  for (uint64_t i = 0; i < n; ++i) {
    float v = vo_dist(X + i * d, C + assign[i] * d, d, metric);
    acc = acc + v;
  }
  return acc;
}
float vo_cost(const float *X, uint64_t n, const float *C, const uint64_t *assign, uint64_t d) {
  return vo_cost_m(X, n, C, assign, d, 0);
}

/* ---- ivfflat.rs:73-100  build_kmeans ------------------------------------- */
/* init_idx replaces initialize_centroids (ivfflat.rs:18-27): the reference
 * draws k indices WITH replacement from an unseeded thread_rng, which is not
 * reproducible by design, so the draw is injected.  Loop: assign, update,
 * bitwise compare (to_hashkey base.rs:113-117 == f32::to_bits equality),
 * break if equal else centroids=new; final assign with the last centroids.
 * iters_run = number of loop bodies executed (including the one that broke). */
int vo_kmeans_m(const float *X, uint64_t n, uint64_t d, uint64_t k, uint64_t max_iterations,
                const uint64_t *init_idx, float *C, uint64_t *assign, uint64_t *iters_run, int metric) {
  for (uint64_t c = 0; c < k; ++c) memcpy(C + c * d, X + init_idx[c] * d, d * sizeof(float));
  float *Cn = (float *)malloc((k * d + 1) * sizeof(float));
  uint64_t it = 0;
  for (uint64_t i = 0; i < max_iterations; ++i) {
    int rc = vo_assign_m(X, n, C, k, d, assign, metric);
    if (rc != VO_OK) {
      free(Cn);
      return rc;
    }
    vo_update(X, n, assign, k, d, Cn);
    ++it;
    if (memcmp(C, Cn, k * d * sizeof(float)) == 0) break;
    memcpy(C, Cn, k * d * sizeof(float));
  }
  free(Cn);
  if (iters_run) *iters_run = it;
  return vo_assign_m(X, n, C, k, d, assign, metric);
}
int vo_kmeans(const float *X, uint64_t n, uint64_t d, uint64_t k, uint64_t max_iterations,
              const uint64_t *init_idx, float *C, uint64_t *assign, uint64_t *iters_run) {
  return vo_kmeans_m(X, n, d, k, max_iterations, init_idx, C, assign, iters_run, 0);
}

/* ---- ivfflat.rs:102-136  build_index (k-means part) ---------------------- */
/* best of num_attempts by strict cost < best_cost starting from +inf; with
 * num_attempts == 0 (or no cost < inf, e.g. NaN/inf cost) nothing is kept:
 * *kept = 0 and the index has EMPTY centroids/assignments.  init_idx holds
 * num_attempts * k injected draws.  The inverted lists (ivfflat.rs:123-127)
 * are ids[c] = ascending vec_ids with assign == c; callers derive them. */
int vo_build_m(const float *X, uint64_t n, uint64_t d, uint64_t k, uint64_t num_attempts,
               uint64_t max_iterations, const uint64_t *init_idx, float *C, uint64_t *assign,
               float *best_cost_out, int *kept, uint64_t *best_attempt, int metric) {
  float best = INFINITY;
  *kept = 0;
  float *Ct = (float *)malloc((k * d + 1) * sizeof(float));
  uint64_t *at = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
  for (uint64_t a = 0; a < num_attempts; ++a) {
    int rc = vo_kmeans_m(X, n, d, k, max_iterations, init_idx + a * k, Ct, at, NULL, metric);
    if (rc != VO_OK) {
      free(Ct);
      free(at);
      return rc;
    }
    float cost = vo_cost_m(X, n, Ct, at, d, metric);
    if (cost < best) {
      best = cost;
      memcpy(C, Ct, k * d * sizeof(float));
      memcpy(assign, at, n * sizeof(uint64_t));
      *kept = 1;
      if (best_attempt) *best_attempt = a;
    }
  }
  free(Ct);
  free(at);
  *best_cost_out = best;
  return VO_OK;
}
int vo_build(const float *X, uint64_t n, uint64_t d, uint64_t k, uint64_t num_attempts,
             uint64_t max_iterations, const uint64_t *init_idx, float *C, uint64_t *assign,
             float *best_cost_out, int *kept, uint64_t *best_attempt) {
  return vo_build_m(X, n, d, k, num_attempts, max_iterations, init_idx, C, assign, best_cost_out, kept, best_attempt, 0);
}

/* ---- the index as five flat arrays (ivfflat.rs:8-15) ---------------------
 * values[n*d], centroids[k*d], assignments[n], and the inverted lists in CSR
 * form: list_off[k+1], list_ids[n] (ids[c] = list_ids[list_off[c]..list_off[c+1]]). */

/* ---- ivfflat.rs:153-198  search_approximate ------------------------------ */
int64_t vo_search_m(const float *values, const float *centroids, uint64_t k, uint64_t d,
                    const uint64_t *list_off, const uint64_t *list_ids, const float *query,
                    uint64_t top_k, uint64_t *out_ids, float *out_dist, int metric) {
  /* :155-161 all centroid distances, centroid.squared_euclidean(&query), stable sort */
  vo_pair *cent = (vo_pair *)malloc((k ? k : 1) * sizeof(vo_pair));
  for (uint64_t c = 0; c < k; ++c) {
    cent[c].id = c;
    cent[c].dist = vo_dist(centroids + c * d, query, d, metric);
  }
  int rc = stable_sort_by_dist(cent, k);
  if (rc != VO_OK) {
    free(cent);
    return rc;
  }
  uint64_t n_out = 0, curr = 0, remainder = top_k;
  while (n_out < top_k) { /* :168 */
    if (curr >= k) {       /* :169 index out of bounds -> panic */
      free(cent);
      return VO_ERR_INSUFFICIENT;
    }
    uint64_t c = cent[curr].id;
    uint64_t len = list_off[c + 1] - list_off[c];
    const uint64_t *ids = list_ids + list_off[c];
    vo_pair *p = (vo_pair *)malloc((len ? len : 1) * sizeof(vo_pair));
    for (uint64_t t = 0; t < len; ++t) { /* :172-175 */
      p[t].id = ids[t];
      p[t].dist = vo_dist(values + ids[t] * d, query, d, metric);
    }
    rc = stable_sort_by_dist(p, len); /* :176 */
    if (rc != VO_OK) {
      free(p);
      free(cent);
      return rc;
    }
    uint64_t plen = len < top_k ? len : top_k; /* :177 take(top_k) */
    if (plen < remainder) {                     /* :181-185 */
      for (uint64_t t = 0; t < plen; ++t) {
        out_ids[n_out] = p[t].id;
        out_dist[n_out++] = p[t].dist;
      }
      remainder -= plen;
      curr += 1;
      free(p);
    } else { /* :186-194 both remaining branches take `remainder` items and stop */
      for (uint64_t t = 0; t < remainder; ++t) {
        out_ids[n_out] = p[t].id;
        out_dist[n_out++] = p[t].dist;
      }
      free(p);
      break;
    }
  }
  free(cent);
  return (int64_t)n_out;
}
int64_t vo_search(const float *values, const float *centroids, uint64_t k, uint64_t d,
                  const uint64_t *list_off, const uint64_t *list_ids, const float *query,
                  uint64_t top_k, uint64_t *out_ids, float *out_dist) {
  return vo_search_m(values, centroids, k, d, list_off, list_ids, query, top_k, out_ids, out_dist, 0);
}

/* ---- extension (NOT in the reference): nprobe search ----------------------
 * BASELINE.json cfg3/cfg4 name nprobe=32.  Definition:
 * score all rows of the P nearest lists (P = min(nprobe, k)), ONE global
 * stable sort by distance over the concatenation in probe-rank order (so ties
 * break by probe rank, then position in the list), take top_k.  With P=1 and
 * |list| >= top_k this equals vo_search. */
int64_t vo_search_nprobe_m(const float *values, const float *centroids, uint64_t k, uint64_t d,
                           const uint64_t *list_off, const uint64_t *list_ids,
                           const float *query, uint64_t top_k, uint64_t nprobe,
                           uint64_t *out_ids, float *out_dist, int metric) {
  vo_pair *cent = (vo_pair *)malloc((k ? k : 1) * sizeof(vo_pair));
  for (uint64_t c = 0; c < k; ++c) {
    cent[c].id = c;
    cent[c].dist = vo_dist(centroids + c * d, query, d, metric);
  }
  int rc = stable_sort_by_dist(cent, k);
  if (rc != VO_OK) {
    free(cent);
    return rc;
  }
  uint64_t P = nprobe < k ? nprobe : k, total = 0;
  for (uint64_t j = 0; j < P; ++j) total += list_off[cent[j].id + 1] - list_off[cent[j].id];
  vo_pair *p = (vo_pair *)malloc((total ? total : 1) * sizeof(vo_pair));
  uint64_t o = 0;
  for (uint64_t j = 0; j < P; ++j) {
    uint64_t c = cent[j].id;
    for (uint64_t t = list_off[c]; t < list_off[c + 1]; ++t) {
      p[o].id = list_ids[t];
      p[o++].dist = vo_dist(values + list_ids[t] * d, query, d, metric);
    }
  }
  free(cent);
  rc = stable_sort_by_dist(p, total);
  if (rc != VO_OK) {
    free(p);
    return rc;
  }
  uint64_t m = top_k < total ? top_k : total;
  for (uint64_t i = 0; i < m; ++i) {
    out_ids[i] = p[i].id;
    out_dist[i] = p[i].dist;
  }
  free(p);
  return (int64_t)m;
}
int64_t vo_search_nprobe(const float *values, const float *centroids, uint64_t k, uint64_t d,
                         const uint64_t *list_off, const uint64_t *list_ids,
                         const float *query, uint64_t top_k, uint64_t nprobe,
                         uint64_t *out_ids, float *out_dist) {
  return vo_search_nprobe_m(values, centroids, k, d, list_off, list_ids, query, top_k, nprobe, out_ids, out_dist, 0);
}

/* ---- ivfflat.rs:200-213  add (the centroid choice) ------------------------ */
/* min_by over (i, centroid.squared_euclidean(&embedding)), first minimum;
 * the caller's vec_id is ignored by the reference (:209). NaN -> panic when
 * k >= 2; k == 0 -> unwrap on None. Returns the chosen cluster or an error. */
int64_t vo_add_cluster_m(const float *centroids, uint64_t k, uint64_t d, const float *x, int metric) {
  if (k == 0) return VO_ERR_EMPTY;
  uint64_t best = 0;
  float bd = vo_dist(centroids, x, d, metric);
  if (k >= 2 && isnan(bd)) return VO_ERR_NAN;
  for (uint64_t c = 1; c < k; ++c) {
    float dc = vo_dist(centroids + c * d, x, d, metric);
    if (isnan(dc)) return VO_ERR_NAN;
    if (dc < bd) {
      bd = dc;
      best = c;
    }
  }
  return (int64_t)best;
}
int64_t vo_add_cluster(const float *centroids, uint64_t k, uint64_t d, const float *x) {
  return vo_add_cluster_m(centroids, k, d, x, 0);
}
