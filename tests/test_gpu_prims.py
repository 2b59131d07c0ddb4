"""The device primitives of csrc/hc_prims.hip (stable radix sort, exclusive sum, ordered selection, unique) against numpy:
every key / value shape the library sorts, sizes around the tile (2 048) and the workgroup-range boundaries, partial bit
ranges, heavy duplicates (stability), all-equal keys, and sizes in the millions."""
import ctypes as C

import numpy as np
import pytest

from haploconduct_amd import _native as N

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 63, 64, 65, 2047, 2048, 2049, 4096, 10000, 2048 * 1024 + 7, 3000001]


def _sort(keys, vals, begin, end):
    ko, vo = np.empty_like(keys), (np.empty_like(vals) if vals is not None else None)
    N.check(N.lib.hc_dev_radix_sort(keys.itemsize, vals.itemsize if vals is not None else 0, keys.ctypes.data,
                                    vals.ctypes.data if vals is not None else None, ko.ctypes.data, vo.ctypes.data if vals is not None else None,
                                    keys.size, begin, end), "hc_dev_radix_sort")
    return ko, vo


@pytest.mark.parametrize("kdt,vdt", [(np.uint32, np.uint32), (np.uint64, np.uint32), (np.uint64, np.uint64), (np.uint64, None)])
@pytest.mark.parametrize("n", SIZES)
def test_radix_sort_is_numpys_stable_sort(kdt, vdt, n):
    rng = np.random.default_rng(n * 7 + np.dtype(kdt).itemsize)
    bits = 8 * np.dtype(kdt).itemsize
    for shape in ("random", "few", "narrow", "equal"):
        if shape == "random":
            keys = rng.integers(0, 2 ** bits, n, dtype=kdt, endpoint=False) if bits == 32 else rng.integers(0, 2 ** 63, n, dtype=np.uint64) * 2 + rng.integers(0, 2, n, dtype=np.uint64)
        elif shape == "few":  # seven distinct keys: long runs of equal keys, stability shows
            keys = rng.choice(rng.integers(0, 2 ** 31, 7), n).astype(kdt) << kdt(bits - 32)
        elif shape == "narrow":  # only some middle bits differ
            keys = (rng.integers(0, 1 << 13, n).astype(kdt) << kdt(11)) | kdt(5)
        else:
            keys = np.full(n, 12345, kdt)
        vals = np.arange(n, dtype=vdt) if vdt is not None else None
        for begin, end in ((0, bits), (0, 13), (11, 24), (bits - 9, bits), (5, 5)):
            ko, vo = _sort(keys, vals, begin, end)
            mask = ((1 << (end - begin)) - 1) if end > begin else 0
            field = (keys >> kdt(begin)) & kdt(mask)
            order = np.argsort(field, kind="stable")
            assert np.array_equal(ko, keys[order]), (shape, begin, end)
            if vals is not None:
                assert np.array_equal(vo, vals[order]), (shape, begin, end)
            if n > 100000 and (begin, end) != (0, bits):
                break  # the large sizes: the full range and one partial one


@pytest.mark.parametrize("dt", [np.uint32, np.uint64])
@pytest.mark.parametrize("n", SIZES + [2048 * 1024 * 3 + 1, 2048 * 8192 + 5])
def test_exclusive_sum(dt, n):
    rng = np.random.default_rng(n)
    a = rng.integers(0, 50 if dt == np.uint32 else 1 << 40, n).astype(dt)
    out = np.empty_like(a)
    N.check(N.lib.hc_dev_exclusive_sum(a.itemsize, a.ctypes.data, out.ctypes.data, n), "hc_dev_exclusive_sum")
    want = np.concatenate([np.zeros(1, dt), np.cumsum(a, dtype=dt)[:-1]])
    assert np.array_equal(out, want)


@pytest.mark.parametrize("n", SIZES + [2048 * 8192 + 5])
@pytest.mark.parametrize("density", [0.0, 0.03, 0.5, 1.0])
def test_select_flagged(n, density):
    rng = np.random.default_rng(n + int(density * 100))
    flags = (rng.random(n) < density).astype(np.uint8) * rng.integers(1, 255, n).astype(np.uint8)
    idx = np.empty(n, np.uint32)
    k = C.c_uint64()
    N.check(N.lib.hc_dev_select_flagged(flags.ctypes.data, n, idx.ctypes.data, C.byref(k)), "hc_dev_select_flagged")
    want = np.nonzero(flags)[0].astype(np.uint32)
    assert k.value == want.size and np.array_equal(idx[:k.value], want)


@pytest.mark.parametrize("n", SIZES)
def test_unique(n):
    rng = np.random.default_rng(n)
    a = np.sort(rng.integers(0, max(2, n // 3), n).astype(np.uint64) * np.uint64(0x100000001))
    out = np.empty_like(a)
    k = C.c_uint64()
    N.check(N.lib.hc_dev_unique_u64(a.ctypes.data, n, out.ctypes.data, C.byref(k)), "hc_dev_unique_u64")
    want = np.unique(a)
    assert k.value == want.size and np.array_equal(out[:k.value], want)
