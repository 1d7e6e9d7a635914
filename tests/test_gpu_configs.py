"""The other BASELINE configurations as parity cases: C4 (500-nt ESTs against 2-10 kb windows, the
traceback branch of the ladder) and C5 (one long cDNA with 20+ introns against a wide window: the
recursive linear-space branch, sub-problems of every size)."""
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_align(item):
    from spaln_amd import abi, defaults
    from oracle import host_logic
    w, q, s5, s3, kw = item
    sc = defaults.scoring(**kw)
    ps = abi.ProblemSet()
    p = ps.add(q, w, s5, s3)
    return host_logic.align_s(sc, p)


def _check_corners(skl, q, w):
    flags, cnt = int(skl[0][0]), int(skl[0][1])
    assert flags == 1 and cnt == skl.shape[0] - 1
    c = skl[1:]
    dm, dn = np.diff(c[:, 0]), np.diff(c[:, 1])
    assert (dm >= 0).all() and (dn >= 0).all()
    assert c[0, 0] >= 0 and c[-1, 0] <= len(q) and c[0, 1] >= 0 and c[-1, 1] <= len(w)
    assert ((dm == dn) | (dm == 0) | (dn == 0)).all()
    return c


def test_c4_est_batch():
    """C4 shape: 500-nt fragments of 2 kb transcripts, 1 % error, windows = locus of the fragment +- 1 kb"""
    from spaln_amd import abi, defaults, engine, synth
    rng = np.random.default_rng(synth.SEED + 404)
    n_q = 3000
    items = []
    sc = defaults.scoring()
    ps = abi.ProblemSet()
    for w, q, s5, s3, exons in synth.make_batch(n_q, seed=synth.SEED + 4040, sub=0.01, indel=0.001):
        a0 = int(rng.integers(0, len(q) - 500))
        frag = q[a0:a0 + 500]
        # genomic span of the fragment: cumulative exon lengths -> window coordinates
        pos, lo, hi = 0, None, None
        for e0, e1 in exons:
            L = e1 - e0
            if lo is None and a0 < pos + L:
                lo = e0 + (a0 - pos)
            if a0 + 500 <= pos + L:
                hi = e0 + (a0 + 500 - pos)
                break
            pos += L
        hi = exons[-1][1] if hi is None else hi
        b0, b1 = max(0, lo - 1000), min(len(w), hi + 1000)
        items.append((w[b0:b1], frag, s5[b0:b1 + 1], s3[b0:b1 + 1]))
        ps.add(frag, w[b0:b1], s5[b0:b1 + 1], s3[b0:b1 + 1])
    eng = engine.Engine(0)
    res = eng.align_s(sc, ps)
    # the same batch with the stragglers' linear-space rounds in front of the big forward sweep instead of
    # beside it (two streams) and with a small MaxVmfSpace (a third of the batch in those rounds)
    os.environ["SPDP_OVERLAP"] = "0"
    try:
        res1 = eng.align_s(sc, ps)
        sc_small = defaults.scoring(max_vmf_space=6 * 1024 * 1024)
        res_small1 = eng.align_s(sc_small, ps)
    finally:
        del os.environ["SPDP_OVERLAP"]
    res_small = eng.align_s(sc_small, ps)
    eng.close()
    for x, y in ((res, res1), (res_small, res_small1)):
        for i, ((s0, k0), (s1, k1)) in enumerate(zip(x, y)):
            assert s0 == s1 and k0.tolist() == k1.tolist(), i
    n_full = 0
    for (score, skl), (w, q, _, _) in zip(res, items):
        assert skl.shape[0] >= 3
        c = _check_corners(skl, q, w)
        n_full += (c[-1, 0] - c[0, 0]) >= 450            # (nearly) the whole fragment is aligned
    assert n_full > 0.9 * n_q
    pick = sorted(rng.choice(n_q, size=128, replace=False).tolist())
    with mp.Pool(min(os.cpu_count() or 1, 64)) as pool:
        want = pool.map(_oracle_align, [items[i] + ({},) for i in pick])
    for i, (ws, wskl) in zip(pick, want):
        assert res[i][0] == ws and res[i][1].ravel().tolist() == (wskl or []), i


def test_c5_long_cdna():
    """C5 shape, scaled to what the CPU oracle can check: 6 kb cDNA with 24 exons against a ~45 kb window.
    2 * m * (n + m) is ~20x MaxVmfSpace and the one-level estimate does not fit either, so lspS_ng takes
    the RECURSIVE branch (rcsv_postwork): halves of halves down to tracebacks."""
    from spaln_amd import abi, defaults, engine, synth
    rng = np.random.default_rng(synth.SEED + 505)
    g = synth.make_gene(rng, n_exons=24, mrna_len=6000, flank=1000, intron_lo=500, intron_hi=3000)
    w, q = defaults.encode(g.window), defaults.encode(g.query)
    s5, s3 = synth.splice_signals(g.window)
    kw = dict(max_vmf_space=4 * 1024 * 1024)
    sc = defaults.scoring(**kw)
    ps = abi.ProblemSet()
    ps.add(q, w, s5, s3)
    eng = engine.Engine(0)
    (score, skl), = eng.align_s(sc, ps)
    eng.close()
    c = _check_corners(skl, q, w)
    cols = set(int(x) for x in c[:, 1])
    hits = sum((e0 in cols) + (e1 in cols) for e0, e1 in g.exons)
    assert hits >= 40                                     # 48 exon boundaries planted
    ws, wskl = _oracle_align((w, q, s5, s3, kw))
    assert score == ws and skl.ravel().tolist() == wskl


def test_c5_full_size_properties():
    """C5 at BASELINE size -- one 50 kb cDNA with 25 exons against its ~190 kb locus, default parameters:
    the recursive linear-space branch all the way down (16-wave blocks at the top levels).  Far beyond
    what the CPU oracle finishes in test time, so size-independent properties: a monotone corner list
    covering the whole query, every planted exon boundary among the corners, the same result twice."""
    from spaln_amd import abi, defaults, engine, synth
    rng = np.random.default_rng(synth.SEED + 55)
    g = synth.make_gene(rng, n_exons=25, mrna_len=50000, flank=1000, intron_lo=1000, intron_hi=10000)
    w, q = defaults.encode(g.window), defaults.encode(g.query)
    s5, s3 = synth.splice_signals(g.window)
    sc = defaults.scoring()
    ps = abi.ProblemSet()
    ps.add(q, w, s5, s3)
    eng = engine.Engine(0)
    (score, skl), = eng.align_s(sc, ps)
    (score2, skl2), = eng.align_s(sc, ps)
    eng.close()
    c = _check_corners(skl, q, w)
    assert c[0, 0] == 0 and c[-1, 0] == len(q)
    cols = set(int(x) for x in c[:, 1])
    hits = sum((e0 in cols) + (e1 in cols) for e0, e1 in g.exons)
    assert hits == 50
    assert score == score2 and skl.tolist() == skl2.tolist()
    assert score > 150000                                  # ~4 per matched base at 2 % divergence


def test_cross_cu_pipelines_equal_one_cu():
    """a launch of a few huge problems spreads each over several CUs (progress words in global memory,
    memory-side coherent boundary entries): bit-identical to the one-CU pipelines, run after run"""
    from spaln_amd import abi, defaults, engine, synth
    rng = np.random.default_rng(synth.SEED + 77)
    g = synth.make_gene(rng, n_exons=20, mrna_len=20000, flank=1000, intron_lo=800, intron_hi=6000)
    w, q = defaults.encode(g.window), defaults.encode(g.query)
    s5, s3 = synth.splice_signals(g.window)
    sc = defaults.scoring()
    ps = abi.ProblemSet()
    ps.add(q, w, s5, s3)
    old = os.environ.get("SPDP_CROSS")
    try:
        os.environ["SPDP_CROSS"] = "0"
        eng = engine.Engine(0)
        (ws, wskl), = eng.align_s(sc, ps)
        us0 = eng.wip_udh(sc, ps, 3)
        eng.close()
        os.environ["SPDP_CROSS"] = "1"
        eng = engine.Engine(0)
        for _ in range(4):
            (s, skl), = eng.align_s(sc, ps)
            assert s == ws and skl.tolist() == wskl.tolist()
            us1 = eng.wip_udh(sc, ps, 3)
            assert all(np.array_equal(a, b) for a, b in zip(us0, us1))
        eng.close()
    finally:
        if old is None:
            os.environ.pop("SPDP_CROSS", None)
        else:
            os.environ["SPDP_CROSS"] = old
    c = _check_corners(wskl, q, w)
    cols = set(int(x) for x in c[:, 1])
    assert sum((e0 in cols) + (e1 in cols) for e0, e1 in g.exons) == 40


def test_cross_cu_fallback_when_a_block_is_missing():
    """the start-up barrier of the cross-CU pipelines times out when a block of the group is not resident
    (here: never launched); the blocks then give up and the host repeats the launch on one CU per problem"""
    from spaln_amd import abi, defaults, engine, synth
    rng = np.random.default_rng(synth.SEED + 505)
    g = synth.make_gene(rng, n_exons=24, mrna_len=6000, flank=1000, intron_lo=500, intron_hi=3000)
    w, q = defaults.encode(g.window), defaults.encode(g.query)
    s5, s3 = synth.splice_signals(g.window)
    sc = defaults.scoring()
    ps = abi.ProblemSet()
    ps.add(q, w, s5, s3)
    eng = engine.Engine(0)
    want = eng.wip_udh(sc, ps, 3)
    os.environ["SPDP_CROSS_TEST_SHORT"] = "1"
    try:
        got = eng.wip_udh(sc, ps, 3)
    finally:
        os.environ.pop("SPDP_CROSS_TEST_SHORT", None)
        eng.close()
    assert all(np.array_equal(a, b) for a, b in zip(want, got))
