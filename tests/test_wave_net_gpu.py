"""The lane networks the kernels sort and merge with (scan.hip.h): lane l <- l ^ J on the VALU (DPP inside a row of 16,
v_permlane16_swap / v_permlane32_swap across rows and halves), the bitonic sort and merge built on them and the rank-counting sort,
against numpy on random and adversarial keys (vers_test_wave_net runs them in one wave)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
KEY_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def run(keys):
    from vers_amd import testhooks
    return testhooks.wave_net(keys).reshape(10, 64)


def cases():
    rng = np.random.default_rng(0x5EED)
    yield rng.integers(0, 2**63, size=128, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    yield np.arange(128, dtype=np.uint64)[::-1].copy()                       # descending
    k = rng.integers(0, 2**40, size=128, dtype=np.uint64)
    k[rng.choice(128, 40, replace=False)] = KEY_MAX                          # padded lists
    yield k
    hi = rng.integers(0, 4, size=128, dtype=np.uint64) << np.uint64(32)      # equal distance bits: the rank sort's fallback
    yield hi | np.arange(128, dtype=np.uint64)


@pytest.mark.parametrize("case", range(4))
def test_lane_networks(case):
    keys = list(cases())[case]
    out = run(keys)
    a, b = keys[:64], keys[64:]
    lanes = np.arange(64)
    for j in range(6):
        assert np.array_equal(out[j], a[lanes ^ (1 << j)]), f"lane ^ {1 << j}"
    assert np.array_equal(out[6], a[63 - lanes]), "lane reversal"
    assert np.array_equal(out[7], np.sort(a)), "bitonic sort"
    assert np.array_equal(out[8], np.sort(a)), "rank-counting sort"
    assert np.array_equal(out[9], np.sort(np.concatenate([a, b]))[:64]), "merge of two sorted lists"


def test_wide_lists_sort_and_merge():
    """wide.hip.h: 256 keys over four registers per lane -- sort, merge of two sorted lists, element access -- against numpy"""
    from vers_amd import testhooks
    rng = np.random.default_rng(0x51DE)
    cases = [rng.integers(0, 2**63, size=512, dtype=np.uint64) * np.uint64(2) + np.uint64(1),
             np.arange(512, dtype=np.uint64)[::-1].copy(), np.arange(512, dtype=np.uint64)]
    k = rng.integers(0, 2**40, size=512, dtype=np.uint64); k[rng.choice(512, 300, replace=False)] = KEY_MAX   # padded lists
    cases.append(k)
    k = (rng.integers(0, 4, size=512, dtype=np.uint64) << np.uint64(32)) | np.arange(512, dtype=np.uint64)        # few distinct distances: the rank sort's tie path
    cases.append(k)
    for keys in cases:
        out = testhooks.wide_net(keys)
        a, b = np.sort(keys[:256]), np.sort(keys[256:])
        assert np.array_equal(out[:256], a) and np.array_equal(out[256:512], b)
        m = np.sort(keys)[:256]
        assert np.array_equal(out[512:768], m)
        assert out[768] == m[0] and out[769] == m[77] and out[770] == m[255]
