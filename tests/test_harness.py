"""BASELINE.json config 1 (plumbing): the reference's demo harness sequence (utils.rs:7-66,117-184) on a small
fastText-format fixture.  CPU part: the loader (skip header, hold out "queen" raw, normalise the rest) against
the committed expectations.  GPU part: build -> add -> save -> load -> search with the raw queen vector."""
import os
import zlib

import numpy as np
import pytest

from tests.golden import make_vec_fixture as mv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VEC = os.path.join(ROOT, "tests", "golden", "wiki_like_300d.vec")


@pytest.fixture(scope="module")
def expected():
    return np.load(os.path.join(ROOT, "tests", "golden", "wiki_like_expected.npz"))


def test_loader_matches_reference_contract(expected):
    from vers_amd.harness import load_wiki_vector
    vecs, w2i, i2w, test_embs = load_wiki_vector(VEC, 300)
    assert vecs.shape == (mv.N_WORDS - 1, 300) and "queen" not in w2i and len(i2w) == mv.N_WORDS - 1
    assert [w for w, _ in test_embs] == ["queen"]
    raw = mv.raw_rows()
    assert np.array_equal(test_embs[0][1].view(np.uint32), raw[mv.QUEEN_AT].view(np.uint32))   # held out RAW
    assert np.uint32(zlib.crc32(vecs.tobytes())) == expected["normalized_crc"][0]              # normalised bit-exactly
    assert i2w[mv.QUEEN_AT] == f"w{mv.QUEEN_AT + 1:04d}"                                        # indices skip the hold-out


@pytest.mark.gpu
def test_ivfflat_demo_sequence(expected, tmp_path):
    from vers_amd.harness import load_wiki_vector, test_ivfflat as run_ivfflat
    vecs, w2i, i2w, test_embs = load_wiki_vector(VEC, 300)
    printed, results = run_ivfflat(vecs, w2i, i2w, mv.K, mv.ATTEMPTS, mv.ITERS, test_embs, init_indices=expected["init"],
                                   index_file_name=os.path.join(tmp_path, "ivfflat.index"))
    assert [w for w, _ in printed] == list(expected["result_words"])
    assert np.array_equal(np.array([i for i, _ in results], dtype=np.uint64), expected["result_ids"])
    assert np.array_equal(np.array([d for _, d in results], dtype=np.float32).view(np.uint32), expected["result_dist_bits"])
    assert np.array_equal(np.array([s for _, s in printed], dtype=np.float32).view(np.uint32), expected["result_sqrt_bits"])
    assert w2i["queen"] == mv.N_WORDS - 1 and printed[0][0] == "queen"   # the added (normalised) queen is its raw self's nearest
