"""The library's own stable radix sort and exclusive scan (tinsel_amd/csrc/tn_sort.h: what the device BVH builder uses since round 5 in place of
rocprim) against numpy, on sizes around the kernels' tile edges (2048 keys per sort tile, 2048 ints per scan tile) and on keys that test
STABILITY: the builder sorts (Morton code << 32 | triangle index) by the code's bytes only and relies on equal codes keeping their index order."""
import numpy as np
import pytest

import tinsel_amd

pytestmark = pytest.mark.gpu
SIZES = [1, 2, 63, 64, 2047, 2048, 2049, 4096, 100003, 524288, (1 << 20) + 17]


@pytest.mark.parametrize("n", SIZES)
def test_scan_equals_numpy(n):
    rng = np.random.default_rng(n)
    v = rng.integers(0, 1000, n).astype(np.int32)
    got = tinsel_amd.selftest_scan(v)
    want = np.concatenate(([0], np.cumsum(v[:-1], dtype=np.int64))).astype(np.int32)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", SIZES)
def test_sort_by_the_high_word_is_stable(n):
    """keys as the builder makes them: few distinct codes (many ties), the low word the element's index"""
    rng = np.random.default_rng(1000 + n)
    code = rng.integers(0, max(2, n//7), n).astype(np.uint64)*np.uint64(0x9e3779b1) & np.uint64(0x3fffffff)
    keys = (code << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    got = tinsel_amd.selftest_sort(keys, 32, 64)
    order = np.argsort(keys >> np.uint64(32), kind="stable")
    assert np.array_equal(got, keys[order])
    assert np.array_equal(got, np.sort(keys))           # (unique keys written in index order: stable by the code == sorted by the whole key)


@pytest.mark.parametrize("bits", [(0, 64), (0, 32), (16, 48), (56, 64)])
def test_sort_by_any_byte_range(bits):
    rng = np.random.default_rng(bits[0]*64 + bits[1])
    keys = rng.integers(0, 1 << 63, 300001, dtype=np.uint64)*np.uint64(2) + rng.integers(0, 2, 300001, dtype=np.uint64)
    got = tinsel_amd.selftest_sort(keys, bits[0], bits[1])
    mask = np.uint64((1 << (bits[1] - bits[0])) - 1) if bits[1] - bits[0] < 64 else np.uint64(0xffffffffffffffff)
    order = np.argsort((keys >> np.uint64(bits[0])) & mask, kind="stable")
    assert np.array_equal(got, keys[order])
