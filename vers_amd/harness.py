"""Host-side mirror of the reference's demo harness for IVFFlat (vers/src/utils.rs:7-66,117-184 and the
commented-out call in main.rs:60-68) -- BASELINE.json config 1 ("plumbing").  Same sequence, same quirks:

  load_wiki_vector : skip the header line, whitespace split, f32 parse; the word "queen" is HELD OUT raw
                     (un-normalised); every other row is normalised with base.rs:99-105 arithmetic.
  run_test         : vectors.push(raw queen) ; index.add(normalize(queen), vec_id) ; save_index ;
                     load_index ; search_approximate(RAW queen vector, 10) ; print word + sqrt(dist).

The arithmetic that decides results (build / add / search) runs on the GPU through the C ABI; this file only
does what the reference does on the host around it.
"""
from __future__ import annotations

import os

import numpy as np

from .index import IVFFlatIndex


def _normalize_rows(a: np.ndarray) -> np.ndarray:
    """Vector::normalize (base.rs:95-105): sequential f32 dot, sqrt, true division; |v| < 1e-6 -> unchanged."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    m = np.sqrt(np.add.accumulate((a * a).astype(np.float32), axis=1, dtype=np.float32)[:, -1]).astype(np.float32)
    out = a.copy()
    big = ~(m < np.float32(1e-6))
    out[big] = (a[big] / m[big, None]).astype(np.float32)
    return out


_F32_MAX = float(np.finfo(np.float32).max)
_F32_OVERFLOW_TIE = _F32_MAX + 2.0 ** 103  # the midpoint between f32::MAX and 2^128: decimals at or above it parse to inf


def parse_f32(text: str) -> np.float32:
    """`str::parse::<f32>()` (utils.rs:31-36): the decimal CORRECTLY rounded to f32 (nearest, ties to even).
    float() is correctly rounded to f64; rounding that once more is off by one ulp exactly when the f64 lands on the
    midpoint of two neighbouring f32 while the decimal itself does not (double rounding: "1.00000017881393432617187499"
    is below the midpoint of 1+2^-23 and 1+2^-22, its f64 IS that midpoint and ties-to-even then picks the wrong side).
    Only those inputs take the exact path (a rational comparison); everything else is one float() and one cast."""
    if "_" in text:
        raise ValueError(f"invalid float literal {text!r}")  # Python accepts 1_000, Rust does not (the reference panics on unwrap)
    d = float(text)
    if d != d or d in (float("inf"), float("-inf")):
        return np.float32(d)
    with np.errstate(over="ignore"):
        f = np.float32(d)
    fd = float(f)
    if fd == d:
        return f  # the f64 is an f32: nothing was rounded the second time
    if np.isinf(f):  # |d| beyond f32::MAX's rounding boundary -- or exactly ON it with the decimal a hair below
        if abs(d) != _F32_OVERFLOW_TIE:
            return f
        from decimal import Decimal
        from fractions import Fraction
        exact = abs(Fraction(Decimal(text)))
        return f if exact >= Fraction(_F32_OVERFLOW_TIE) else np.float32(np.copysign(_F32_MAX, d))
    nb = float(np.nextafter(f, np.float32(np.inf if d > fd else -np.inf)))
    if np.isinf(nb):
        nb = float(np.copysign(2.0 ** 128, d))
    if (fd + nb) / 2.0 != d:
        return f  # not a tie in f64: the cast rounded the right way
    from decimal import Decimal
    from fractions import Fraction
    exact, mid = Fraction(Decimal(text)), Fraction(d)
    if exact == mid:
        return f  # a true tie: to even, which is what the cast did
    other = np.float32(nb) if abs(nb) <= _F32_MAX else np.float32(np.copysign(np.inf, d))
    lo, hi = (f, other) if fd < nb else (other, f)
    return hi if exact > mid else lo


def _open_text(file_path: str):
    """plain text like the reference's file, or the same gzipped (the committed fixture): no unpacked copy is left behind"""
    if str(file_path).endswith(".gz"):
        import gzip
        return gzip.open(file_path, "rt", encoding="utf-8")
    return open(file_path, "r", encoding="utf-8")


def load_wiki_vector(file_path: str, d: int):
    """utils.rs:7-66.  Returns (all_vecs [n, d] f32, word_to_idx, idx_to_word, test_embs [(word, raw emb)])."""
    words, rows, test_embs = [], [], []
    with _open_text(file_path) as f:
        next(f)  # header line (utils.rs:26 skip(1))
        for line in f:
            parts = line.split()
            word = parts[0]
            emb = np.array([parse_f32(x) for x in parts[1:]], dtype=np.float32)
            if emb.size != d:
                raise ValueError(f"expected {d} values for {word!r}")  # try_into().unwrap() panics in the reference
            if word == "queen":
                test_embs.append((word, emb))
                continue
            words.append(word)
            rows.append(emb)
    all_vecs = _normalize_rows(np.stack(rows)) if rows else np.zeros((0, d), dtype=np.float32)
    word_to_idx = {w: i for i, w in enumerate(words)}
    idx_to_word = {i: w for i, w in enumerate(words)}
    return all_vecs, word_to_idx, idx_to_word, test_embs


def fmt_f32(x) -> str:
    """Rust's `{}` for an f32: the shortest decimal that round-trips, never in scientific notation."""
    x = np.float32(x)
    if np.isnan(x):
        return "NaN"
    if np.isinf(x):
        return "inf" if x > 0 else "-inf"
    return np.format_float_positional(x, unique=True, trim="-")


def run_test(index: IVFFlatIndex, index_file_name: str, vectors: list, word_to_idx: dict, idx_to_word: dict, test_embs,
             echo=None):
    """utils.rs:117-158 for T = IVFFlatIndex.  `vectors` is a python list of rows (the reference's Vec<Vector<N>>).
    echo: callable that receives the lines the reference println!s (the CLI passes print)."""
    say = echo or (lambda _line: None)
    for word, emb in test_embs:
        vec_id = len(vectors)
        vectors.append(np.asarray(emb, dtype=np.float32).copy())        # raw, un-normalised (utils.rs:129-131)
        say(f"Inserting {word} {vec_id}")                               # utils.rs:133
        idx_to_word[vec_id] = word
        word_to_idx[word] = vec_id
        index.add(_normalize_rows(emb[None])[0], vec_id)                # utils.rs:136
    try:
        index.save_index(index_file_name)                               # utils.rs:140-143
        say("Index saved successfully!")
    except OSError as e:
        import sys
        print(f"Index save failed: {e}", file=sys.stderr)
    reload_index = IVFFlatIndex.load_index(index_file_name, index.d, index.device)   # utils.rs:145
    results = reload_index.search_approximate(vectors[word_to_idx["queen"]], 10)     # utils.rs:148: the RAW vector
    out = [(idx_to_word[i], np.sqrt(np.float32(dist), dtype=np.float32)) for i, dist in results]  # utils.rs:151-157
    for i, (word, dist) in enumerate(out):
        say(f"{i}. Word: {word}. Distance: {fmt_f32(dist)}")
    reload_index.close()
    return out, results


def test_ivfflat(vectors: np.ndarray, word_to_idx: dict, idx_to_word: dict, num_clusters: int, num_attempts: int,
                 max_iterations: int, test_embs, init_indices=None, index_file_name: str = "ivfflat.index", device: int = 0,
                 echo=None):
    """utils.rs:160-184.  `init_indices` injects the reference's unseeded centroid draws (ivfflat.rs:18-27)."""
    if echo:
        echo("IVFFlat Index:-----")                                     # utils.rs:169
    ivfflat = IVFFlatIndex.build_index(num_clusters, num_attempts, max_iterations, vectors, init_indices=init_indices,
                                       device=device)
    vec_list = [v for v in np.asarray(vectors, dtype=np.float32)]
    try:
        return run_test(ivfflat, index_file_name, vec_list, word_to_idx, idx_to_word, test_embs, echo=echo)
    finally:
        ivfflat.close()


def main(argv=None) -> int:
    """The reference's demo (main.rs:54-68 with the IVFFlat call enabled: k=20, attempts=3, iterations=10) as a command:
         python -m vers_amd.harness <file.vec> [num_clusters [num_attempts [max_iterations]]] [--seed S] [--index-file P]
    load -> build_index -> add(normalize(queen)) -> save_index -> load_index -> search_approximate(raw queen, 10) on the
    MI355X, printing the reference's own lines (utils.rs:133,141,151-157,169).  The vector length is read from the
    file's first data line (the reference fixes it at compile time: const DIM = 300).  --seed makes the centroid draws
    of initialize_centroids reproducible (numpy default_rng; the reference draws from an unseeded thread_rng)."""
    import argparse
    import time
    ap = argparse.ArgumentParser(prog="python -m vers_amd.harness", description=main.__doc__)
    ap.add_argument("vec_file")
    ap.add_argument("num_clusters", nargs="?", type=int, default=20)
    ap.add_argument("num_attempts", nargs="?", type=int, default=3)
    ap.add_argument("max_iterations", nargs="?", type=int, default=10)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--index-file", default="ivfflat.index")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    path = a.vec_file
    with _open_text(path) as f:
        next(f)
        d = len(f.readline().split()) - 1
    vecs, w2i, i2w, test_embs = load_wiki_vector(path, d)
    rng = np.random.default_rng(a.seed)
    init = rng.integers(0, max(len(vecs), 1), size=a.num_attempts * a.num_clusters)
    t0 = time.perf_counter()
    test_ivfflat(vecs, w2i, i2w, a.num_clusters, a.num_attempts, a.max_iterations, test_embs, init_indices=init,
                 index_file_name=a.index_file, device=a.device, echo=print)
    print(f"Time taken to test: {time.perf_counter() - t0:.6f}s")        # main.rs:101-102 (Duration's {:?})
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
