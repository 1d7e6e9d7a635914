"""The fiber scheduler of the seeded drivers (spaln_amd/csrc/spdp_seeded_rv.h) on the CPU: thousands of toy walks in flight
on a few worker threads, requests sorted into latency classes with dispatcher lanes of their own, every walk gets exactly
the answers to its own requests -- whatever thread it resumes on, from deep inside a recursion."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle


def _run(n, max_parks, depth, lanes, env):
    lib = C.CDLL(oracle.build_walk_check())
    lib.walk_check_scheduler.argtypes = [C.c_int] * 6 + [C.c_void_p]
    out = np.full(n, -1, dtype=np.int64)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        nb = lib.walk_check_scheduler(n, max_parks, depth, *lanes, out.ctypes.data)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    q = np.arange(n, dtype=np.int64)
    parks = 1 + (q * 7919) % max_parks
    want = parks * q * 131 + 17 * parks * (parks - 1) // 2
    return nb, out, want


@pytest.mark.parametrize("lanes", [(1, 0, 0), (2, 1, 1), (1, 1, 3)])
def test_every_walk_gets_its_own_answers(lanes):
    nb, out, want = _run(5000, 9, 40, lanes, {"SPDP_SEED_THREADS": "6", "SPDP_SEED_WALKS": "1500", "SPDP_SEED_BATCH": "64"})
    assert nb > 0
    assert np.array_equal(out, want)


def test_one_walk_one_thread_and_a_deep_recursion():
    # 300 frames x 1 KB on a 1 MB fiber stack; a single walk never gathers a batch: the idle rule has to fire
    nb, out, want = _run(1, 5, 300, (2, 1, 1), {"SPDP_SEED_THREADS": "1"})
    assert nb == int(1 + (0 * 7919) % 5) and np.array_equal(out, want)


def test_more_walks_than_fibers_in_flight():
    nb, out, want = _run(3000, 4, 5, (2, 1, 1), {"SPDP_SEED_THREADS": "4", "SPDP_SEED_WALKS": "7", "SPDP_SEED_BATCH": "1000"})
    assert np.array_equal(out, want)


def test_out_of_stacks_caps_the_walks_in_flight():
    # mmap "fails" after 5 fibers: the other walks start as those fibers come free; with none at all the call fails
    nb, out, want = _run(400, 4, 5, (2, 1, 1), {"SPDP_SEED_THREADS": "4", "SPDP_SEED_TEST_STACKS": "5"})
    assert nb > 0 and np.array_equal(out, want)
    nb, out, want = _run(400, 4, 5, (2, 1, 1), {"SPDP_SEED_THREADS": "4", "SPDP_SEED_TEST_STACKS": "0"})
    assert nb == -1
