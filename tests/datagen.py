"""Deterministic synthetic vectors shared by tests, golden fixtures and bench.py.

Integer hashing + exactly-representable float conversions only (no libm
transcendental), so the same bits come out on every machine AND out of the HIP
generator kernel in vers_amd/csrc/gen.hip (vers_gen_rows_dev), which restates
this recipe.  SURVEY.md section 8(d) "Synthetic data".

  h(seed,i,j)  = mix64(mix64(seed + i*0xD1342543DE82EF95) + j)           (u64 wraparound)
  noise(i,j)   = f32(sum of the four 16-bit fields of h - 131070) * 2^-16  (Irwin-Hall(4), var 1/3)
  Dist-U row i = normalize(noise(i,:))
  Dist-C row i = normalize(centre[i % n_modes] + sigma * noise(i,:)),  centre m = Dist-U row m of seed_c
normalize = vers base.rs:99-105 arithmetic (sequential f32 dot, sqrt, true division).
"""
from __future__ import annotations

import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_ROW_MUL = np.uint64(0xD1342543DE82EF95)


def mix64(z: np.ndarray) -> np.ndarray:
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def noise(seed: int, rows: np.ndarray, d: int) -> np.ndarray:
    """[len(rows), d] f32 noise for the given global row indices."""
    rows = np.asarray(rows, dtype=np.uint64)
    with np.errstate(over="ignore"):
        rk = mix64(np.uint64(seed) + rows * _ROW_MUL)  # [n]
        h = mix64(rk[:, None] + np.arange(d, dtype=np.uint64)[None, :])
    s = ((h & np.uint64(0xFFFF)) + ((h >> np.uint64(16)) & np.uint64(0xFFFF))
         + ((h >> np.uint64(32)) & np.uint64(0xFFFF)) + (h >> np.uint64(48))).astype(np.int64) - 131070
    return (s.astype(np.float32) * np.float32(2.0 ** -16)).astype(np.float32)


def _seq_dot_rows(a: np.ndarray) -> np.ndarray:
    return np.add.accumulate((a * a).astype(np.float32), axis=1, dtype=np.float32)[:, -1]


def normalize_rows(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    m = np.sqrt(_seq_dot_rows(a)).astype(np.float32)
    out = a.copy()
    big = ~(m < np.float32(1e-6))
    out[big] = (a[big] / m[big, None]).astype(np.float32)
    return out


def dist_u(seed: int, n: int, d: int, start: int = 0) -> np.ndarray:
    return normalize_rows(noise(seed, np.arange(start, start + n), d))


def dist_c(seed: int, n: int, d: int, n_modes: int, sigma: float, seed_c: int | None = None,
           start: int = 0) -> np.ndarray:
    seed_c = (seed ^ 0xC0FFEE) if seed_c is None else seed_c
    rows = np.arange(start, start + n, dtype=np.uint64)
    centres = dist_u(seed_c, n_modes, d)
    nz = (np.float32(sigma) * noise(seed, rows, d)).astype(np.float32)
    raw = (centres[(rows % np.uint64(n_modes)).astype(np.int64)] + nz).astype(np.float32)
    return normalize_rows(raw)


def default_sigma(d: int) -> np.float32:
    """noise norm ~ 0.5: sigma * sqrt(d/3) = 0.5."""
    return np.float32(0.5 / np.sqrt(d / 3.0))
