"""BASELINE.json config 1 (plumbing): the reference's demo harness sequence (utils.rs:7-66,117-184) on a small
fastText-format fixture.  CPU part: the loader (skip header, hold out "queen" raw, normalise the rest) against
the committed expectations.  GPU part: build -> add -> save -> load -> search with the raw queen vector."""
import os
import zlib

import numpy as np
import pytest

from tests.golden import make_vec_fixture as mv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VEC_GZ = os.path.join(ROOT, "tests", "golden", "wiki_like_300d.vec.gz")


@pytest.fixture(scope="module")
def VEC(tmp_path_factory):
    """the committed fixture is gzipped (2,000 words x 300: 1.2 MB instead of 6 MB); the loader reads plain text"""
    import gzip
    p = os.path.join(tmp_path_factory.mktemp("vec"), "wiki_like_300d.vec")
    with gzip.open(VEC_GZ, "rb") as src, open(p, "wb") as dst:
        dst.write(src.read())
    return p


@pytest.fixture(scope="module")
def expected():
    return np.load(os.path.join(ROOT, "tests", "golden", "wiki_like_expected.npz"))


def test_loader_matches_reference_contract(expected, VEC):
    from vers_amd.harness import load_wiki_vector
    vecs, w2i, i2w, test_embs = load_wiki_vector(VEC, 300)
    assert vecs.shape == (mv.N_WORDS - 1, 300) and "queen" not in w2i and len(i2w) == mv.N_WORDS - 1
    assert [w for w, _ in test_embs] == ["queen"]
    raw = mv.raw_rows()
    assert np.array_equal(test_embs[0][1].view(np.uint32), raw[mv.QUEEN_AT].view(np.uint32))   # held out RAW
    assert np.uint32(zlib.crc32(vecs.tobytes())) == expected["normalized_crc"][0]              # normalised bit-exactly
    assert i2w[mv.QUEEN_AT] == f"w{mv.QUEEN_AT + 1:04d}"                                        # indices skip the hold-out


@pytest.mark.gpu
def test_ivfflat_demo_sequence(expected, tmp_path, VEC):
    from vers_amd.harness import load_wiki_vector, test_ivfflat as run_ivfflat
    vecs, w2i, i2w, test_embs = load_wiki_vector(VEC, 300)
    printed, results = run_ivfflat(vecs, w2i, i2w, mv.K, mv.ATTEMPTS, mv.ITERS, test_embs, init_indices=expected["init"],
                                   index_file_name=os.path.join(tmp_path, "ivfflat.index"))
    assert [w for w, _ in printed] == list(expected["result_words"])
    assert np.array_equal(np.array([i for i, _ in results], dtype=np.uint64), expected["result_ids"])
    assert np.array_equal(np.array([d for _, d in results], dtype=np.float32).view(np.uint32), expected["result_dist_bits"])
    assert np.array_equal(np.array([s for _, s in printed], dtype=np.float32).view(np.uint32), expected["result_sqrt_bits"])
    assert w2i["queen"] == mv.N_WORDS - 1 and printed[0][0] == "queen"   # the added (normalised) queen is its raw self's nearest


def test_rust_display_of_f32():
    """`Distance: {}` (utils.rs:155) prints an f32 the way Rust's Display does: shortest round-trip, positional"""
    from vers_amd.harness import fmt_f32
    assert fmt_f32(np.float32(1.0)) == "1" and fmt_f32(np.float32(0.5)) == "0.5" and fmt_f32(np.float32(0.1)) == "0.1"
    assert fmt_f32(np.float32(1e-7)) == "0.0000001" and fmt_f32(np.float32(16777216.0)) == "16777216"
    assert fmt_f32(np.float32(0.0)) == "0" and fmt_f32(np.sqrt(np.float32(2.0))) == "1.4142135"


@pytest.mark.gpu
def test_cli_prints_the_reference_lines(expected, tmp_path):
    """python -m vers_amd.harness <file.vec.gz> 20 2 10 --seed S: the demo end to end on the GPU, lines as the reference
    prints them; ids / distances of the result lines == the oracle run with the same (seeded) draws."""
    import subprocess
    import sys
    from oracle import c_oracle as co
    from vers_amd.harness import fmt_f32, load_wiki_vector
    import gzip
    plain = os.path.join(tmp_path, "w.vec")
    with gzip.open(VEC_GZ, "rb") as src, open(plain, "wb") as dst:
        dst.write(src.read())
    r = subprocess.run([sys.executable, "-m", "vers_amd.harness", VEC_GZ, str(mv.K), "2", str(mv.ITERS), "--seed", "7", "--index-file",
                        os.path.join(tmp_path, "ivfflat.index")], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0] == "IVFFlat Index:-----" and lines[1] == f"Inserting queen {mv.N_WORDS - 1}" and lines[2] == "Index saved successfully!"
    vecs, w2i, i2w, test_embs = load_wiki_vector(plain, 300)
    init = np.random.default_rng(7).integers(0, len(vecs), size=2 * mv.K)
    b = co.build_index(vecs, mv.K, 2, mv.ITERS, init)
    qn = co.normalize(test_embs[0][1][None])[0]
    c = co.add_cluster(b["centroids"], qn)
    ids = [list(l) for l in b["ids"]]; ids[c].append(len(vecs))
    values = np.concatenate([vecs, qn[None]], axis=0)
    ri, rd = co.search_approximate(values, b["centroids"], ids, test_embs[0][1], 10)
    i2w[len(vecs)] = "queen"
    want = [f"{i}. Word: {i2w[int(v)]}. Distance: {fmt_f32(np.sqrt(np.float32(dd), dtype=np.float32))}" for i, (v, dd) in enumerate(zip(ri, rd))]
    assert lines[3:3 + len(want)] == want
    assert lines[3 + len(want)].startswith("Time taken to test: ")
    assert sorted(os.listdir(tmp_path)) == ["ivfflat.index", "w.vec"]   # the gzipped input is read in place: no stray unpacked copy


def test_parse_f32_is_correctly_rounded():
    """utils.rs:31-36 `parse::<f32>()` rounds the DECIMAL once; going through an f64 rounds twice.  Witnesses: decimals a
    hair off the midpoint of two f32 whose f64 is exactly that midpoint (the cast then ties to even, the wrong way)."""
    from decimal import Decimal, getcontext
    from fractions import Fraction
    from vers_amd.harness import parse_f32
    w = "1.00000017881393432617187499"                     # just below the midpoint of 1+2^-23 and 1+2^-22
    assert parse_f32(w).view(np.uint32) == 0x3F800001 and np.float32(float(w)).view(np.uint32) == 0x3F800002
    assert parse_f32("1.000000178813934326171875").view(np.uint32) == 0x3F800002      # the exact tie: to even
    assert parse_f32("1.00000017881393432617187501").view(np.uint32) == 0x3F800002
    assert parse_f32("-1.00000017881393432617187499").view(np.uint32) == 0xBF800001
    getcontext().prec = 400
    rng = np.random.default_rng(5)
    for bits in rng.integers(1, 0x7F000000, size=300):     # subnormal and normal f32, midpoint to the next one up
        f = np.uint32(bits).view(np.float32)
        nb = np.nextafter(f, np.float32(np.inf))
        mid = (Fraction(float(f)) + Fraction(float(nb))) / 2
        md = Decimal(mid.numerator) / Decimal(mid.denominator)
        eps = Decimal(10) ** (md.adjusted() - 60)
        even, odd = (f, nb) if (int(bits) & 1) == 0 else (nb, f)
        assert parse_f32(str(md)).view(np.uint32) == even.view(np.uint32)
        assert parse_f32(str(md + eps)).view(np.uint32) == nb.view(np.uint32)
        assert parse_f32(str(md - eps)).view(np.uint32) == f.view(np.uint32)
    # the overflow boundary (f32::MAX + half an ulp) and plain values
    assert parse_f32("3.40282356779733661637539395458142568447e38") == np.finfo(np.float32).max
    assert np.isinf(parse_f32("3.40282356779733661637539395458142568448e38")) and np.isinf(parse_f32("1e39"))
    assert parse_f32("0.1").view(np.uint32) == np.float32(0.1).view(np.uint32) and parse_f32("-0.0").view(np.uint32) == 0x80000000
    assert np.isnan(parse_f32("NaN")) and np.isinf(parse_f32("-inf"))
    with pytest.raises(ValueError):
        parse_f32("1_0.5")                                  # Python's float() takes it, Rust's parse does not


def test_loader_reads_the_gzipped_fixture_in_place(expected, tmp_path):
    """no unpacked copy is written anywhere (the CLI used to leave `<index stem>.vec` next to the index file)"""
    from vers_amd.harness import load_wiki_vector
    before = set(os.listdir(tmp_path))
    vecs, w2i, i2w, test_embs = load_wiki_vector(VEC_GZ, 300)
    assert np.uint32(zlib.crc32(vecs.tobytes())) == expected["normalized_crc"][0] and set(os.listdir(tmp_path)) == before
