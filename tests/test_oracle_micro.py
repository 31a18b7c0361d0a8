"""Hand-computable micro-cases pinning every quirk of the reference path.

Both restatements (C and NumPy) are run on each case; expected values are worked out by hand
from the Rust source (paths relative to /root/reference/vers/src)."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import np_oracle as no

f32 = np.float32
ORACLES = [pytest.param(co, id="c"), pytest.param(no, id="numpy")]


def panics(o):
    return (co.OraclePanic, no.RefPanic)


@pytest.mark.parametrize("o", ORACLES)
def test_squared_euclidean_is_sequential_f32(o):
    # base.rs:119-126: 1e8^2 = 1e16 absorbs every following +1 when summed left to right;
    # a pairwise / f64 sum would give 1e16 + 3.
    a = np.array([1e8, 1, 1, 1], dtype=f32)
    z = np.zeros(4, dtype=f32)
    assert f32(o.squared_euclidean(a, z)) == f32(1e16)
    # order matters: small terms first survive
    assert f32(o.squared_euclidean(a[::-1].copy(), z)) == f32(f32(3.0) + f32(1e16))
    assert f32(o.squared_euclidean(np.array([1, 2, 3], f32), np.zeros(3, f32))) == f32(14.0)
    # bit-symmetric in its arguments
    x = np.array([0.1, -0.7, 0.33], f32); y = np.array([0.9, 0.2, -0.5], f32)
    assert f32(o.squared_euclidean(x, y)).tobytes() == f32(o.squared_euclidean(y, x)).tobytes()


@pytest.mark.parametrize("o", ORACLES)
def test_normalize_small_magnitude_returns_input(o):
    # base.rs:99-105: magnitude < 1e-6 -> unchanged; else true division by sqrt(dot)
    tiny = np.array([[1e-7, 0, 0]], dtype=f32)
    assert np.array_equal(o.normalize(tiny), tiny)
    v = np.array([[3, 4, 0]], dtype=f32)
    assert np.array_equal(o.normalize(v), np.array([[f32(3) / f32(5), f32(4) / f32(5), 0]], dtype=f32))


@pytest.mark.parametrize("o", ORACLES)
def test_assign_tie_goes_to_lowest_centroid(o):
    # ivfflat.rs:33-43 min_by keeps the first of equal minima
    X = np.array([[0, 0], [2, 0]], dtype=f32)
    C = np.array([[1, 0], [-1, 0], [0, 1], [1, 0]], dtype=f32)
    assert list(o.assign_to_clusters(X, C)) == [0, 0]


@pytest.mark.parametrize("o", ORACLES)
def test_update_empty_cluster_is_zero_and_mean_is_sum_over_count(o):
    # ivfflat.rs:47-71
    X = np.array([[1, 2], [3, 6], [5, 1]], dtype=f32)
    a = np.array([2, 2, 0], dtype=np.uint64)
    Cn = o.update_centroids(X, a, 3)
    assert np.array_equal(Cn, np.array([[5, 1], [0, 0], [2, 4]], dtype=f32))


@pytest.mark.parametrize("o", ORACLES)
def test_update_sum_order_is_data_order(o):
    # sums[c] = ((0 + x0) + x1) + x2 in ascending index order (ivfflat.rs:52-55)
    X = np.array([[1e8], [1], [-1e8], [1]], dtype=f32)
    a = np.zeros(4, dtype=np.uint64)
    # ((0+1e8)+1) = 1e8 ; + -1e8 = 0 ; + 1 = 1 ; / 4
    assert o.update_centroids(X, a, 1)[0, 0] == f32(0.25)


@pytest.mark.parametrize("o", ORACLES)
def test_cost_is_sequential_fold(o):
    X = np.array([[1e4], [1], [1]], dtype=f32)  # D = 1e8, 1, 1 against centroid 0
    C = np.zeros((1, 1), dtype=f32)
    assert f32(o.kmeans_cost(X, C, np.zeros(3, dtype=np.uint64))) == f32(1e8)


@pytest.mark.parametrize("o", ORACLES)
def test_kmeans_converges_and_breaks_on_bitwise_equal(o):
    # two tight blobs; after one update the centroids stop changing -> break at iteration 2
    X = np.array([[0, 0], [0, 1], [10, 0], [10, 1]], dtype=f32)
    C, a, iters = o.build_kmeans(X, 2, 50, np.array([0, 2], dtype=np.uint64))
    assert list(a) == [0, 0, 1, 1]
    assert np.array_equal(C, np.array([[0, 0.5], [10, 0.5]], dtype=f32))
    assert iters == 2
    # max_iterations = 0: initial centroids + one assign (ivfflat.rs:98)
    C0, a0, it0 = o.build_kmeans(X, 2, 0, np.array([1, 1], dtype=np.uint64))
    assert it0 == 0 and np.array_equal(C0, X[[1, 1]]) and list(a0) == [0, 0, 0, 0]


@pytest.mark.parametrize("o", ORACLES)
def test_duplicate_init_centroids_leave_second_empty(o):
    X = np.array([[0, 0], [1, 0], [5, 5]], dtype=f32)
    C, a, _ = o.build_kmeans(X, 2, 1, np.array([0, 0], dtype=np.uint64))
    # all points go to centroid 0 (tie -> lowest), centroid 1 becomes the zero vector
    assert np.array_equal(C[1], np.zeros(2, dtype=f32))
    assert np.array_equal(C[0], np.array([f32(6) / f32(3), f32(5) / f32(3)], dtype=f32))


@pytest.mark.parametrize("o", ORACLES)
def test_build_index_keeps_first_best_and_zero_attempts_keeps_nothing(o):
    X = np.array([[0, 0], [0, 1], [10, 0], [10, 1]], dtype=f32)
    init = np.array([0, 2, 2, 0], dtype=np.uint64)  # both attempts converge to the same cost
    b = o.build_index(X, 2, 2, 10, init)
    assert b["kept"] and np.array_equal(b["centroids"], np.array([[0, 0.5], [10, 0.5]], dtype=f32))  # attempt 0 kept (strict <)
    assert [list(l) for l in b["ids"]] == [[0, 1], [2, 3]]
    b0 = o.build_index(X, 2, 0, 10, np.zeros(0, dtype=np.uint64))
    assert not b0["kept"] and b0["centroids"].shape[0] == 0


def _tiny_index():
    values = np.array([[0, 0], [0, 2], [10, 0], [10, 3], [10, 1]], dtype=f32)
    centroids = np.array([[0, 1], [10, 1]], dtype=f32)
    ids = [[0, 1], [2, 3, 4]]
    return values, centroids, ids


@pytest.mark.parametrize("o", ORACLES)
def test_search_spill_is_concatenation_not_merge(o):
    # ivfflat.rs:166-195: nearest list gives its 2 rows, the next list its own top-(4-2)
    values, centroids, ids = _tiny_index()
    q = np.array([1, 0], dtype=f32)
    i, d = o.search_approximate(values, centroids, ids, q, 4)
    assert list(i) == [0, 1, 2, 4]
    assert list(d) == [1.0, 5.0, 81.0, 82.0]
    # top_k smaller than the first list: only that list, truncated
    i, d = o.search_approximate(values, centroids, ids, q, 1)
    assert list(i) == [0] and list(d) == [1.0]
    # exact fit takes the third branch (:191-194)
    i, _ = o.search_approximate(values, centroids, ids, q, 2)
    assert list(i) == [0, 1]
    # top_k = 0 -> empty, no panic
    i, _ = o.search_approximate(values, centroids, ids, q, 0)
    assert len(i) == 0


@pytest.mark.parametrize("o", ORACLES)
def test_search_stable_ties_by_list_position(o):
    values = np.array([[1, 0], [-1, 0], [0, 1], [0, -1]], dtype=f32)
    centroids = np.array([[0, 0]], dtype=f32)
    ids = [[3, 1, 0, 2]]  # list order, not id order, breaks ties (appended rows come last)
    i, d = o.search_approximate(values, centroids, ids, np.zeros(2, dtype=f32), 3)
    assert list(i) == [3, 1, 0] and list(d) == [1.0, 1.0, 1.0]


@pytest.mark.parametrize("o", ORACLES)
def test_search_too_few_vectors_panics(o):
    values, centroids, ids = _tiny_index()
    with pytest.raises(panics(o)):
        o.search_approximate(values, centroids, ids, np.zeros(2, dtype=f32), 6)  # ivfflat.rs:169 OOB


@pytest.mark.parametrize("o", ORACLES)
def test_empty_lists_are_skipped(o):
    values = np.array([[5, 5], [6, 6]], dtype=f32)
    centroids = np.array([[0, 0], [5, 5]], dtype=f32)
    ids = [[], [0, 1]]
    i, _ = o.search_approximate(values, centroids, ids, np.zeros(2, dtype=f32), 2)
    assert list(i) == [0, 1]


@pytest.mark.parametrize("o", ORACLES)
def test_nan_panics_when_compared(o):
    values, centroids, ids = _tiny_index()
    q = np.array([np.nan, 0], dtype=f32)
    with pytest.raises(panics(o)):
        o.search_approximate(values, centroids, ids, q, 1)
    with pytest.raises(panics(o)):
        o.search_exhaustive(values, q, 1)
    with pytest.raises(panics(o)):
        o.assign_to_clusters(values, np.array([[np.nan, 0], [0, 0]], dtype=f32))
    with pytest.raises(panics(o)):
        o.add_cluster(np.array([[np.nan, 0], [0, 0]], dtype=f32), q)


@pytest.mark.parametrize("o", ORACLES)
def test_add_picks_first_minimum(o):
    C = np.array([[1, 0], [-1, 0], [0, 1]], dtype=f32)
    assert o.add_cluster(C, np.zeros(2, dtype=f32)) == 0
    assert o.add_cluster(C, np.array([-0.5, 0], dtype=f32)) == 1
    with pytest.raises(panics(o)):
        o.add_cluster(np.zeros((0, 2), dtype=f32), np.zeros(2, dtype=f32))  # ivfflat.rs:207 unwrap on None


@pytest.mark.parametrize("o", ORACLES)
def test_exhaustive_ties_by_index_and_cosine(o):
    X = np.array([[1, 0], [0, 1], [1, 0], [-1, 0]], dtype=f32)
    i, d = o.search_exhaustive(X, np.array([1, 0], dtype=f32), 3)
    assert list(i) == [0, 2, 1] and list(d) == [0.0, 0.0, 2.0]
    i, d = o.search_exhaustive(X, np.array([1, 0], dtype=f32), 4, 1)  # 1 - dot (base.rs:153-155)
    assert list(i) == [0, 2, 1, 3] and list(d) == [0.0, 0.0, 1.0, 2.0]
    i, _ = o.search_exhaustive(X, np.array([1, 0], dtype=f32), 10)  # top_k > n -> n results
    assert len(i) == 4


@pytest.mark.parametrize("o", ORACLES)
def test_nprobe_extension_matches_reference_when_one_list_suffices(o):
    values, centroids, ids = _tiny_index()
    q = np.array([9, 0], dtype=f32)
    a = o.search_approximate(values, centroids, ids, q, 3)
    b = o.search_nprobe(values, centroids, ids, q, 3, 1)
    assert list(a[0]) == list(b[0]) and list(a[1]) == list(b[1])
    i, d = o.search_nprobe(values, centroids, ids, q, 5, 2)  # global order over both lists
    assert list(i) == [2, 4, 3, 0, 1]
