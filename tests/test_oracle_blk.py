"""The block-search restatement (oracle/spdp_oracle_blk.c) against the reference's own recorded runs.

tests/golden/blk_k1.spdg / blk_k3.spdg come from the compiled reference with a recorder on SrchBlk::findblock / TestOutput /
FindHsp (tests/golden/make_blk_goldens.py, oracle/ref_build/blk_tap.cc): an index the reference's own `spaln -W` built
(contiguous 8-mers; five spaced patterns), mixed queries, the vote's state at every TestOutput call and the block pairs
TestOutput handed to FindHsp.  The oracle has to reproduce every record bit for bit."""
import os

import numpy as np
import pytest

from oracle import blk
from tests import spdg

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = ["blk_k1", "blk_k3"]


@pytest.fixture(scope="module", params=FIXTURES)
def case(request):
    fx = spdg.load(os.path.join(HERE, "golden", request.param + ".spdg"))
    ix, keep = blk.index_of(fx)
    return request.param, fx, ix, keep, blk.parse_log(fx)


def test_fixture_covers_the_branches(case):
    name, fx, ix, _, qs = case
    assert ix.kk == (1 if name == "blk_k1" else 3)
    calls = [len(q["calls"]) for q in qs]
    assert max(calls) >= 4 and min(calls) == 1            # single-call queries and queries that go on after "nothing found"
    lens = [q["right"] - q["left"] for q in qs]
    assert min(lens) < 130 and max(lens) > 1800
    # both strands win somewhere: rvs of the best pair
    rvs = {int(c[1][2 + 8]) for q in qs for c in q["calls"][:1] if c[1] is not None}
    assert rvs == {0, 1}


def test_vote_state_at_every_testoutput_call(case):
    _, _, ix, _, qs = case
    n = 0
    for qi, q in enumerate(qs):
        for ci, (want_vote, _) in enumerate(q["calls"]):
            got = blk.vote(ix, q["codes"], q["left"], q["right"], ci)
            assert got is not None, (qi, ci)
            assert np.array_equal(got[0], want_vote), (qi, ci)
            n += 1
    assert n >= 50


def test_block_pairs_handed_to_findhsp(case):
    _, _, ix, _, qs = case
    n = 0
    for qi, q in enumerate(qs):
        for ci, (_, want_pairs) in enumerate(q["calls"]):
            if want_pairs is None:                        # TestOutput called FindHsp for no pair at that call
                continue
            got = blk.vote(ix, q["codes"], q["left"], q["right"], ci)[1]
            k = int(got[1])
            assert k >= 1
            assert np.array_equal(got[2:2 + 9 * k], want_pairs[2:2 + 9 * k]), (qi, ci)
            n += 1
    assert n >= 20


def test_findblock_ends_where_the_reference_ended(case):
    """a query whose recorded run made fewer than MinSigpr + 1 calls with all of them 'nothing found' must have ended by
    itself: asking the oracle for one more call gives none (the scan met in the middle)"""
    _, _, ix, _, qs = case
    for q in qs:
        if len(q["calls"]) == ix.minsigpr + 1:
            assert blk.vote(ix, q["codes"], q["left"], q["right"], len(q["calls"])) is None


# ---- Dhash::resize: a position hash of the queues grows -------------------------------------------------------------------
@pytest.fixture(scope="module")
def grow_case():
    base = spdg.load(os.path.join(HERE, "golden", "blk_k3.spdg"))
    fx = spdg.load(os.path.join(HERE, "golden", "blk_k3_grow.spdg"))
    ix, keep = blk.index_of(base)
    single = blk.parse_log(dict(q_log=fx["q_log"], blk_prm=base["blk_prm"]))
    batch = blk.parse_log(dict(q_log=fx["q_log_batch"], blk_prm=base["blk_prm"]))
    return ix, keep, single, batch


def test_growth_in_a_fresh_process(grow_case):
    """each query run by the reference in a process of its own: the oracle equals every call, and tables do grow on the way"""
    ix, _, single, _ = grow_case
    g0 = blk.grows()
    for qi, q in enumerate(single):
        for ci, (want_vote, want_pairs) in enumerate(q["calls"]):
            got = blk.vote(ix, q["codes"], q["left"], q["right"], ci)
            assert got is not None and np.array_equal(got[0], want_vote), (qi, ci)
            if want_pairs is not None:
                k = int(got[1][1])
                assert np.array_equal(got[1][2:2 + 9 * k], want_pairs[2:2 + 9 * k]), (qi, ci)
    assert blk.grows() - g0 >= 3


def test_grown_tables_persist_on_the_reference_worker(grow_case):
    """the same queries in ONE reference process: later queries see the position hashes at the size earlier ones grew them to.
    With that memory carried along the oracle equals the batch run as well; without it, it must differ somewhere (the
    fixture would otherwise not witness the dependence)."""
    ix, _, single, batch = grow_case
    assert [q["codes"].tolist() for q in single] == [q["codes"].tolist() for q in batch]
    carry = blk.new_carry(ix)
    differs = 0
    for qi, q in enumerate(batch):
        last = len(q["calls"]) - 1
        for ci, (want_vote, _) in enumerate(q["calls"]):
            c2 = carry.copy()
            rec = blk.vote_carry(ix, q["codes"], q["left"], q["right"], ci, c2)
            assert rec is not None and np.array_equal(rec[:want_vote.size], want_vote), (qi, ci)
            fresh = blk.vote(ix, q["codes"], q["left"], q["right"], ci)
            differs += not np.array_equal(fresh[0], want_vote)
            if ci == last:
                carry = c2
    assert differs > 0
