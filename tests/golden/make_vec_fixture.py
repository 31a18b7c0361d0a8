"""Writes tests/golden/wiki_like_300d.vec.gz (a 2,000-word stand-in for fastText wiki-news-300d-1M.vec, which the
reference fetches from the network and does not bundle) and tests/golden/wiki_like_expected.npz: what the
reference harness sequence (utils.rs:7-66,117-184; main.rs:60-68 parameters k=20, iters=10, attempts
reduced to 2 with injected draws) must print for it, computed by the C oracle and cross-checked against the
NumPy oracle.  Values are multiples of 2^-10 written with 10 decimals, so every f32 parser reads them exactly.

Run from the repo root:  python tests/golden/make_vec_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle as co  # noqa: E402
from oracle import np_oracle as no  # noqa: E402
from tests import datagen as dg  # noqa: E402
from tests.golden.make_golden import bits, init_draws, same  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
N_WORDS, D, K, ATTEMPTS, ITERS = 2000, 300, 20, 2, 10   # ~2,000 words x 300, contains "queen"
QUEEN_AT = 1137


def raw_rows():
    x = dg.dist_c(0x51C0, N_WORDS, D, 60, dg.default_sigma(D))
    scale = (1.0 + (np.arange(N_WORDS) % 7)[:, None] * 0.25).astype(np.float32)   # un-normalised, like fastText
    return (np.round(x * scale * 1024.0) / 1024.0).astype(np.float32)


def words():
    return [("queen" if i == QUEEN_AT else f"w{i:04d}") for i in range(N_WORDS)]


def expected():
    raw = raw_rows(); ws = words()
    keep = [i for i in range(N_WORDS) if i != QUEEN_AT]
    X = co.normalize(raw[keep])
    same(X, no.normalize(raw[keep]), "normalize")
    init = init_draws(0xCAFE, ATTEMPTS, K, len(keep))
    b = co.build_index(X, K, ATTEMPTS, ITERS, init)
    bn = no.build_index(X, K, ATTEMPTS, ITERS, init)
    same(b["centroids"], bn["centroids"], "centroids"); same(b["assignments"], bn["assignments"], "assign")
    queen_raw = raw[QUEEN_AT]
    qn = co.normalize(queen_raw[None])[0]
    c = co.add_cluster(b["centroids"], qn)
    ids = [list(l) for l in b["ids"]]; ids[c].append(len(keep))
    values = np.concatenate([X, qn[None]], axis=0)
    ri, rd = co.search_approximate(values, b["centroids"], ids, queen_raw, 10)      # RAW query (utils.rs:148)
    ni, nd = no.search_approximate(values, b["centroids"], ids, queen_raw, 10)
    same(ri, ni, "search ids"); same(rd, nd, "search dist")
    idx_to_word = {j: ws[i] for j, i in enumerate(keep)}; idx_to_word[len(keep)] = "queen"
    return dict(init=init, centroids_bits=bits(b["centroids"]), assignments=b["assignments"], add_cluster=np.array([c], np.uint64),
                result_ids=ri, result_dist_bits=bits(rd), result_sqrt_bits=bits(np.sqrt(rd, dtype=np.float32)),
                result_words=np.array([idx_to_word[int(i)] for i in ri]), normalized_crc=np.array([np.uint32(__import__("zlib").crc32(X.tobytes()))]))


if __name__ == "__main__":
    raw = raw_rows(); ws = words()
    import gzip
    # gzip with mtime 0: the committed file is reproducible; tests gunzip it into a temp dir (the reference's loader
    # reads plain text).  Values are multiples of 2^-10: at most 10 decimals, trailing zeros dropped.
    with gzip.GzipFile(os.path.join(OUT, "wiki_like_300d.vec.gz"), "wb", compresslevel=9, mtime=0) as f:
        f.write(f"{N_WORDS} {D}\n".encode())
        for w, r in zip(ws, raw):
            f.write((w + " " + " ".join((f"{v:.10f}".rstrip("0").rstrip(".") or "0") for v in r) + "\n").encode())
    np.savez_compressed(os.path.join(OUT, "wiki_like_expected.npz"), **expected())
    print("wrote wiki_like_300d.vec.gz", os.path.getsize(os.path.join(OUT, "wiki_like_300d.vec.gz")), "bytes")
