"""CPU tier: compression_amd/csrc/sort_order.h (the order libstdc++'s std::sort gives tied keys, which
decides which of two equal-penalty symbols PmfToQuantizedCdf adjusts, pmf_to_cdf_kernels.cc:179,196)
compiled for the host and compared with std::sort itself."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def check(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("native") / "sort_order_check.so")
    subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", so,
                    os.path.join(HERE, "native", "sort_order_check.cc")], check=True)
    fn = C.CDLL(so).sort_order_check
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]

    def run(keys, descending):
        keys = np.ascontiguousarray(keys, np.float64)
        mine = np.zeros(len(keys), np.uint32)
        theirs = np.zeros(len(keys), np.uint32)
        same = fn(keys.ctypes.data, len(keys), int(descending), mine.ctypes.data, theirs.ctypes.data)
        return bool(same), mine, theirs
    return run


def killer(n):
    """Median-of-three adversary (Musser): drives the quicksort part into its depth limit, so the
    heap-sort branch runs."""
    k = n // 2
    a = np.zeros(n)
    for i in range(1, k + 1):
        if i % 2:
            a[i - 1] = i
            a[i] = k + i
        a[k + i - 1] = 2 * i
    return a


def test_same_permutation_as_std_sort(check):
    rng = np.random.default_rng(0)
    cases = []
    for n in (0, 1, 2, 3, 15, 16, 17, 18, 31, 33, 64, 100, 257, 1000, 4097):
        cases.append(rng.random(n))                                   # no ties
        cases.append(rng.integers(0, 3, n).astype(float))             # almost all ties
        cases.append(rng.integers(0, max(n // 4, 1), n).astype(float))
        cases.append(np.zeros(n))                                     # one key
        x = np.abs(np.arange(n) - (n - 1) / 2)
        cases.append(np.exp(-0.5 * (x / max(n / 8, 1)) ** 2))         # symmetric table: tied pairs
        cases.append(np.where(x > n / 3, np.inf, x))                  # infinities (count-1 symbols)
        cases.append(np.sort(rng.random(n)))
        cases.append(np.sort(rng.random(n))[::-1])
    for n in (64, 512, 2048, 6000):
        cases.append(killer(n))
        cases.append(-killer(n))
    for keys in cases:
        for descending in (False, True):
            same, mine, theirs = check(keys, descending)
            assert same, (len(keys), descending, mine[:20], theirs[:20])


def test_ties_are_not_stable(check):
    """Guards the premise: if std::sort were stable on these, sort_order.h would be unnecessary."""
    x = np.abs(np.arange(201) - 100.0)
    _, mine, _ = check(x, False)
    stable = np.argsort(x, kind="stable")
    assert not (mine == stable).all()
