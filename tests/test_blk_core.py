"""The product's block-vote routine without a GPU: spaln_amd/csrc/spdp_blk_core.h -- the text the device kernel is compiled
from -- built with the host compiler (oracle/blk_check.cpp) and held against the reference's recorded runs
(tests/golden/blk_*.spdg) and against the independent C restatement (oracle/spdp_oracle_blk.c) on random queries."""
import os

import numpy as np
import pytest

from oracle import blk
from tests import spdg

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module", params=["blk_k1", "blk_k3", "blk_p1"])
def case(request):
    fx = spdg.load(os.path.join(HERE, "golden", request.param + ".spdg"))
    ix, keep = blk.index_of(fx)
    _IX[0] = ix
    return fx, ix, keep, blk.parse_log(fx)


_IX = [None]


def RUNS(runs, pairs):
    return [sorted(x) for x in blk.runs_near_pairs(_IX[0], runs, pairs)]


def same(got, want):
    if not got["reached"] or not np.array_equal(got["head"], want["head"]):
        return False
    if not all(np.array_equal(a, b) for a, b in zip(got["qb"], want["qb"])):
        return False
    if got["runs"] != RUNS(want["runs"], got["pairs"]):          # (the product reports the run scores around its pairs)
        return False
    if want["pairs"] is not None:
        k = len(got["pairs"])
        return k >= 1 and np.array_equal(got["pairs"], want["pairs"][:k])
    return True


@pytest.mark.parametrize("touched_cap", [4096, 8])          # 8: the list of touched slots overflows, the full clean-up runs
def test_core_equals_recorded_reference_runs(case, touched_cap):
    _, ix, _, qs = case
    for qi, q in enumerate(qs):
        for ci, (vote, pairs) in enumerate(q["calls"]):
            got = blk.core_vote(ix, q["codes"], q["left"], q["right"], ci, touched_cap=touched_cap)
            assert same(got, blk.split_recorded(vote, pairs)), (qi, ci)


def test_core_equals_oracle_on_random_queries(case):
    fx, ix, _, qs = case
    rng = np.random.default_rng(4711)
    pool = [q["codes"] for q in qs]
    n = 0
    for t in range(60):
        a = pool[int(rng.integers(len(pool)))]
        lo = int(rng.integers(0, max(1, len(a) - 40)))
        b = a[lo:lo + int(rng.integers(30, 900))].copy()
        hits = rng.random(b.size) < rng.choice([0.0, 0.03, 0.15])
        if int(fx["blk_prm"][blk.PRM["drna"]]):
            b[hits] = rng.choice(np.array([2, 3, 5, 9, 16], dtype=np.uint8), size=int(hits.sum()))   # A C G T N (N: not a residue)
        else:
            b[hits] = rng.choice(np.array(list(range(3, 23)) + [2], dtype=np.uint8), size=int(hits.sum()))   # amino acids, X
        left = int(rng.integers(0, 5)); right = len(b) - int(rng.integers(0, 5))
        for stop in (0, 1):
            want = blk.vote(ix, b, left, right, stop)
            got = blk.core_vote(ix, b, left, right, stop)
            if want is None:
                assert not got["reached"], (t, stop)
                continue
            w = blk.split_recorded(want[0], want[1])
            w["pairs"] = want[1][2:].reshape(-1, 9)
            assert same(got, w) and len(got["pairs"]) == len(w["pairs"]), (t, stop)
            n += 1
    assert n >= 40


def test_core_grows_its_tables_like_a_fresh_reference_process():
    base = spdg.load(os.path.join(HERE, "golden", "blk_k3.spdg"))
    fx = spdg.load(os.path.join(HERE, "golden", "blk_k3_grow.spdg"))
    ix, _keep = blk.index_of(base)
    _IX[0] = ix
    for qi, q in enumerate(blk.parse_log(dict(q_log=fx["q_log"], blk_prm=base["blk_prm"]))):
        for ci, (vote, pairs) in enumerate(q["calls"]):
            got = blk.core_vote(ix, q["codes"], q["left"], q["right"], ci)
            assert not got["flags"] & 4 and same(got, blk.split_recorded(vote, pairs)), (qi, ci)
