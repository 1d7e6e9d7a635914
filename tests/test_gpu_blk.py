"""Block search on the device (spdp_blk_vote, SURVEY 8 row f4 first slice) against the reference's recorded runs and the oracle."""
import os

import numpy as np
import pytest

from oracle import blk as oblk
from spaln_amd import blocks, engine
from tests import spdg

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def eng():
    e = engine.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module", params=["blk_k1", "blk_k3", "blk_p1"])      # blk_p1: protein queries, the translated index (-KP)
def case(request, eng):
    fx = spdg.load(os.path.join(HERE, "golden", request.param + ".spdg"))
    ix, keep = oblk.index_of(fx)
    dix = blocks.BlockIndex(eng, fx)
    _IX[0] = ix
    yield fx, ix, keep, oblk.parse_log(fx), dix
    dix.free()


_IX = [None]


def RUNS(runs, pairs):
    return [sorted(x) for x in oblk.runs_near_pairs(_IX[0], runs, pairs)]


def same(got, want, exact_pairs=False):
    if not got["reached"] or not np.array_equal(got["head"], want["head"]):
        return False
    if not all(np.array_equal(a, b) for a, b in zip(got["qb"], want["qb"])):
        return False
    if got["runs"] != RUNS(want["runs"], got["pairs"]):          # (the product reports the run scores around its pairs)
        return False
    if want["pairs"] is not None:
        k = len(got["pairs"])
        if exact_pairs and k != len(want["pairs"]):
            return False
        return k >= 1 and np.array_equal(got["pairs"], want["pairs"][:k])
    return True


def test_device_equals_recorded_reference_runs(case):
    """every TestOutput call the reference made, one batch: query i asked for call c = one entry with stop_at = c"""
    _, _, _, qs, dix = case
    queries, ranges, stops, wants = [], [], [], []
    for q in qs:
        for ci, (vote, pairs) in enumerate(q["calls"]):
            queries.append(q["codes"]); ranges.append((q["left"], q["right"])); stops.append(ci)
            wants.append(oblk.split_recorded(vote, pairs))
    out, _ = dix.vote(queries, ranges, stops, out_cap=1 << 14)
    for i, w in enumerate(wants):
        assert same(blocks.split_record(out[i]), w), i


def test_findblock_end_is_reported(case):
    fx, ix, _, qs, dix = case
    sel = [q for q in qs if len(q["calls"]) == ix.minsigpr + 1]
    if not sel and not int(fx["blk_prm"][oblk.PRM["drna"]]):
        pytest.skip("no query of the protein fixture runs out of TestOutput calls")
    assert sel
    out, _ = dix.vote([q["codes"] for q in sel], [(q["left"], q["right"]) for q in sel], [len(q["calls"]) for q in sel])
    for i in range(len(sel)):
        r = blocks.split_record(out[i])
        assert not r["reached"] and r["calls"] == ix.minsigpr + 1


def test_device_equals_oracle_on_a_random_batch(case):
    """2000 fragments (exact, mutated, with Ns, pure noise), more queries than lanes of a small launch reuse their slabs"""
    fx, ix, _, qs, dix = case
    letters = np.array([2, 3, 5, 9, 16] if int(fx["blk_prm"][oblk.PRM["drna"]]) else list(range(3, 23)) + [2], dtype=np.uint8)
    rng = np.random.default_rng(99)
    pool = [q["codes"] for q in qs]
    queries, ranges = [], []
    for t in range(2000):
        a = pool[int(rng.integers(len(pool)))]
        lo = int(rng.integers(0, max(1, len(a) - 40)))
        b = a[lo:lo + int(rng.integers(30, 700))].copy()
        hits = rng.random(b.size) < rng.choice([0.0, 0.02, 0.2, 1.0])
        b[hits] = rng.choice(letters, size=int(hits.sum()))
        queries.append(b); ranges.append((int(rng.integers(0, 4)), len(b) - int(rng.integers(0, 4))))
    for stop in (0, 2):
        os.environ["SPDP_BLK_WAVES_PER_CU"] = "1" if stop else "16"
        out, _ = dix.vote(queries, ranges, [stop] * len(queries), out_cap=1 << 13)
        n = 0
        for i, b in enumerate(queries):
            want = oblk.vote(ix, b, ranges[i][0], ranges[i][1], stop)
            got = blocks.split_record(out[i])
            assert not got["flags"] & (blocks.CUT | blocks.TABLE), i
            if want is None:
                assert not got["reached"], (i, stop)
                continue
            w = oblk.split_recorded(want[0], want[1]); w["pairs"] = want[1][2:].reshape(-1, 9)
            assert same(got, w, exact_pairs=True), (i, stop)
            n += 1
        assert n > (300 if int(fx["blk_prm"][oblk.PRM["drna"]]) else 150)       # (fewer protein fragments reach a TestOutput call)
    os.environ.pop("SPDP_BLK_WAVES_PER_CU", None)


def test_small_record_is_cut_not_overrun(case):
    _, _, _, qs, dix = case
    q = max(qs, key=lambda x: len(x["codes"]))
    out, _ = dix.vote([q["codes"]], [(q["left"], q["right"])], [0], out_cap=40)
    assert int(out[0, 2]) & blocks.CUT and int(out[0, 0]) == 40


def test_device_grows_its_tables_like_a_fresh_reference_process(eng):
    base = spdg.load(os.path.join(HERE, "golden", "blk_k3.spdg"))
    fx = spdg.load(os.path.join(HERE, "golden", "blk_k3_grow.spdg"))
    dix = blocks.BlockIndex(eng, base)
    _IX[0] = oblk.index_of(base)[0]
    _keep = oblk.index_of(base)
    _IX[0] = _keep[0]
    qs = oblk.parse_log(dict(q_log=fx["q_log"], blk_prm=base["blk_prm"]))
    queries, ranges, stops, wants = [], [], [], []
    for q in qs:
        for ci, (vote, pairs) in enumerate(q["calls"]):
            queries.append(q["codes"]); ranges.append((q["left"], q["right"])); stops.append(ci)
            wants.append(oblk.split_recorded(vote, pairs))
    out, _ = dix.vote(queries, ranges, stops, out_cap=1 << 14)
    for i, w in enumerate(wants):
        got = blocks.split_record(out[i])
        assert not got["flags"] & blocks.TABLE and same(got, w), i
    dix.free()


def test_index_read_from_the_reference_file_votes_like_the_recorded_one(eng):
    """the whole of the slice end to end: the library reads the reference's own .bkn, uploads it, votes -- and equals the
    reference's recorded runs"""
    fx = spdg.load(os.path.join(HERE, "golden", "blk_k3.spdg"))
    prm = np.asarray(fx["blk_prm"])
    from_file = blocks.read_index_file(eng.lib, os.path.join(HERE, "golden", "blk_k3.bkn"), ext_block=int(prm[26]), max_out=int(prm[39]))
    dix = blocks.BlockIndex(eng, from_file)
    _IX[0] = oblk.index_of(fx)[0]
    _keep = oblk.index_of(fx)
    _IX[0] = _keep[0]
    qs = oblk.parse_log(fx)
    queries, ranges, stops, wants = [], [], [], []
    for q in qs:
        for ci, (vote, pairs) in enumerate(q["calls"]):
            queries.append(q["codes"]); ranges.append((q["left"], q["right"])); stops.append(ci)
            wants.append(oblk.split_recorded(vote, pairs))
    out, _ = dix.vote(queries, ranges, stops, out_cap=1 << 14)
    for i, w in enumerate(wants):
        assert same(blocks.split_record(out[i]), w), i
    dix.free()
