"""spdp_corner_list (the library's stdskl / stdskl3, spdp_host.cpp) against the oracle's line-by-line restatements
of src/gaps.cc:140-227 on randomised record sets: monotone paths with diagonal, gap and mixed steps, repeats,
steps back, frame shifts.  Host-only entry: runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

from oracle import host_logic, host_logic_h
from spaln_amd import abi, engine


def _run(lib, recs, unit):
    n = len(recs)
    arr = (abi.Skl * max(n, 1))(*[abi.Skl(m, k) for m, k in recs])
    out = (abi.Skl * (2 * n + 2))()
    lib.spdp_corner_list.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    got = lib.spdp_corner_list(arr, n, unit, out)
    assert got >= 0
    return [(out[i].m, out[i].n) for i in range(got)]


def _random_path(rng, unit):
    m, n = int(rng.integers(0, 5)), int(rng.integers(0, 9))
    recs = [(m, n)]
    for _ in range(int(rng.integers(0, 14))):
        kind = int(rng.integers(0, 7))
        dm = int(rng.integers(1, 6)); dn = int(rng.integers(1, 16))
        if kind == 0: m, n = m + dm, n + dm * unit                    # diagonal
        elif kind == 1: n += dn                                      # gap along n
        elif kind == 2: m += dm                                      # gap along m
        elif kind == 3: m, n = m + dm, n + dm * unit + dn            # diagonal + n gap
        elif kind == 4: m, n = m + dm + int(rng.integers(1, 4)), n + dm * unit   # diagonal + m gap
        elif kind == 5: pass                                         # repeat
        else: n = max(0, n - int(rng.integers(1, 4))); m += int(rng.integers(0, 2))   # inconsistent step
        recs.append((m, n))
    order = rng.permutation(len(recs))
    return [recs[i] for i in order]


@pytest.mark.parametrize("unit", [1, 3])
def test_corner_list_equals_restatement(unit):
    lib = engine.load_library()
    rng = np.random.default_rng(100 + unit)
    ref = host_logic.std_skl if unit == 1 else host_logic_h.std_skl3
    for _ in range(3000):
        recs = _random_path(rng, unit)
        want = [tuple(x) for x in ref([list(r) for r in recs])]
        assert _run(lib, recs, unit) == want, recs


def test_corner_list_small_cases():
    lib = engine.load_library()
    assert _run(lib, [], 1) == []
    assert _run(lib, [(3, 4)], 1) == [(3, 4)]
    assert _run(lib, [(0, 0), (5, 5), (9, 9)], 1) == [(0, 0), (9, 9)]
    assert _run(lib, [(0, 0), (5, 9)], 1) == [(0, 0), (5, 5), (5, 9)]
