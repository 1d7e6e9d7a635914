"""The -A0 wavefront kernels with the 64-row tiles of a problem running as a pipeline of waves
(spdp_rowwave<., true> / spdp_rowwave_udh<true>, the default whenever a problem has two tiles or more):
same records, cpos rows and scores as one wave per problem (SPDP_A0_PIPE=0), as the oracle, and as the
relaunch that follows a wave giving up its wait (test hook SPDP_A0_PIPE_TEST_STALL)."""
import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi, synth

pytestmark = pytest.mark.gpu

S_FILES = golden_files("s1_")


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


def _subranges(fx, n, seed, rows=(70, 1400)):
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + seed)
    extra = dict(cano5=fx["cano5"], cano3=fx["cano3"],
                 dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
    ps = abi.ProblemSet()
    for i in range(n):
        m = int(rng.integers(rows[0], min(rows[1], q["a_right"]) + 1))
        al = int(rng.integers(0, q["a_right"] - m + 1))
        bl = int(rng.integers(0, min(800, q["b_right"] - m - 350)))
        br = int(rng.integers(max(bl + m + 300, q["b_right"] - 1500), q["b_right"] + 1))
        exg = (1, 1, 1, 1) if i % 3 == 0 else tuple(int(x) for x in rng.integers(0, 2, size=4))
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, al + m, bl, br, exg, **extra)
    return ps


def _run_all(eng, sc, ps, n_im, intvl):
    fwd = [(s, skl.tolist()) for s, skl in eng.scalar_forward(sc, ps)]
    sco = eng.scalar_scorealone(sc, ps).tolist()
    scores, cpos, ranges, flags = eng.scalar_udh(sc, ps, n_im, intvl)
    return fwd, sco, scores.tolist(), cpos.tolist(), ranges.tolist(), flags.tolist()


@pytest.mark.parametrize("name,m", [("s1_1400nt", 700), ("s1_local", 400)])
def test_pipelined_equals_one_wave(eng, monkeypatch, name, m):
    f = [f for f in S_FILES if f.endswith(name + ".spdg")]
    if not f:
        pytest.skip("fixture not present")
    fx = spdg.load(f[0])
    sc = spdg.scoring(fx, scalar_engines=1)
    n_im = 4
    intvl = (m + n_im) // (n_im + 1)
    # hirschbergS_ng wants one interval for the batch: fixed height for the linear-space leg
    ps = _subranges(fx, 24, 301, rows=(m, m))
    monkeypatch.setenv("SPDP_A0_PIPE", "0")
    want = _run_all(eng, sc, ps, n_im, intvl)
    monkeypatch.delenv("SPDP_A0_PIPE")
    got = _run_all(eng, sc, ps, n_im, intvl)
    for k, (w, g) in enumerate(zip(want, got)):
        assert w == g, k
    monkeypatch.setenv("SPDP_A0_PIPE_TEST_STALL", "1")       # every pipelined launch is repeated one wave per problem
    again = _run_all(eng, sc, ps, n_im, intvl)
    monkeypatch.delenv("SPDP_A0_PIPE_TEST_STALL")
    for k, (w, g) in enumerate(zip(want, again)):
        assert w == g, k
    # ragged heights (1 .. 22 tiles) through the two engines that take them
    ps = _subranges(fx, 48, 302, rows=(70, 2 * m))
    monkeypatch.setenv("SPDP_A0_PIPE", "0")
    want = [(s, skl.tolist()) for s, skl in eng.scalar_forward(sc, ps)], eng.scalar_scorealone(sc, ps).tolist()
    monkeypatch.delenv("SPDP_A0_PIPE")
    got = [(s, skl.tolist()) for s, skl in eng.scalar_forward(sc, ps)], eng.scalar_scorealone(sc, ps).tolist()
    assert want == got


def test_pipelined_against_oracle(eng):
    """tall sub-ranges (up to 22 tiles): records, scores and cpos rows against the oracle"""
    from oracle import oracle
    fx = spdg.load([f for f in S_FILES if f.endswith("s1_1400nt.spdg")][0])
    sc = spdg.scoring(fx, scalar_engines=1)
    ps = _subranges(fx, 10, 303, rows=(900, 1400))
    bad = []
    for i, (p, (s, skl)) in enumerate(zip(ps.items, eng.scalar_forward(sc, ps))):
        ws, wskl = oracle.scalar_forward(sc, p)
        if s != ws or skl.tolist() != wskl.tolist():
            bad.append((i, s, ws, skl.tolist()[:6], wskl.tolist()[:6]))
    assert not bad, bad[:3]
    m, n_im = 1100, 9
    intvl = (m + n_im) // (n_im + 1)
    ps = _subranges(fx, 8, 304, rows=(m, m))
    scores, cpos, ranges, flags = eng.scalar_udh(sc, ps, n_im, intvl)
    for i, p in enumerate(ps.items):
        ws, wcpos, wrng, wflag = oracle.scalar_udh(sc, p, n_im, intvl)
        ok = int(flags[i]) == wflag
        if wflag == 0:
            ok = ok and int(scores[i]) == ws and ranges[i].tolist() == wrng.tolist() and cpos[i].tolist() == wcpos.tolist()
        if not ok:
            bad.append((i, int(scores[i]), ws, int(flags[i]), wflag))
    assert not bad, bad[:3]


# ---- aa x genome: spdh_rowwave<., true> ------------------------------------------------------------
def _subranges_h(fx, n, seed, rows):
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + seed)
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    ps = abi.ProblemSetH()
    for i in range(n):
        m = int(rng.integers(rows[0], min(rows[1], q["a_right"]) + 1))
        al = int(rng.integers(0, q["a_right"] - m + 1))
        bl = int(rng.integers(1, 500))
        br = int(rng.integers(max(bl + 3 * m + 300, q["b_right"] - 1200), q["b_right"] + 1))
        exg = (1, 1, 1, 1) if i % 3 == 0 else tuple(int(x) for x in rng.integers(0, 3 if i % 4 == 1 else 2, size=4))
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
               fx["phs5"], fx["phs3"], al, al + m, bl, br, exg, exin=(q["b_left"], q["b_right"]), dinc=dinc)
    return ps


def _run_all_h(eng, sc, ps, n_im, intvl):
    fwd = [(s, skl.tolist()) for s, skl in eng.scalar_forward_h(sc, ps)]
    sco = [s for s, _ in eng.scalar_forward_h(sc, ps, traceback=False)]
    out = [fwd, sco]
    if n_im:
        scores, cpos, ranges, flags = eng.scalar_udh_h(sc, ps, n_im, intvl)
        out += [flags.tolist(), [(int(s), c.tolist(), r.tolist()) for s, c, r, f in zip(scores, cpos, ranges, flags) if f == 0]]
    return out


@pytest.mark.parametrize("name,local", [("h1_400aa", False), ("h1_local", True)])
def test_pipelined_equals_one_wave_h(eng, monkeypatch, name, local):
    f = [f for f in golden_files("h1_") if f.endswith(name + ".spdg")]
    if not f:
        pytest.skip("fixture not present")
    fx = spdg.load(f[0])
    assert bool(fx["prm"]["local"]) == local
    sc = spdg.scoring_h(fx, scalar_engines=1)
    m = min(300, fx["prm"]["a_right"])
    n_im = 3
    intvl = (m + n_im) // (n_im + 1)
    ps = _subranges_h(fx, 20, 311, rows=(m, m))
    monkeypatch.setenv("SPDP_A0_PIPE", "0")
    want = _run_all_h(eng, sc, ps, n_im, intvl)
    monkeypatch.setenv("SPDP_A0_PIPE", "1")
    got = _run_all_h(eng, sc, ps, n_im, intvl)
    for k, (w, g) in enumerate(zip(want, got)):
        assert w == g, k
    monkeypatch.setenv("SPDP_A0_PIPE_TEST_STALL", "1")
    again = _run_all_h(eng, sc, ps, n_im, intvl)
    monkeypatch.delenv("SPDP_A0_PIPE_TEST_STALL")
    for k, (w, g) in enumerate(zip(want, again)):
        assert w == g, k
    # ragged heights
    ps = _subranges_h(fx, 40, 312, rows=(66, 400))
    monkeypatch.setenv("SPDP_A0_PIPE", "0")
    want = _run_all_h(eng, sc, ps, 0, 0)
    monkeypatch.setenv("SPDP_A0_PIPE", "1")
    got = _run_all_h(eng, sc, ps, 0, 0)
    assert want == got


# ---- -A1: spdp_exact<., true>, the 16-row stripes of a problem as a pipeline ---------------------------------
def test_a1_pipelined_equals_one_group(eng, monkeypatch):
    """alignS_ng / HomScoreS_ng under -A1 (scalar_engines = 2) on every taller cDNA fixture with its own ladder
    parameters (traceback, linear-space and local branches): pipelined stripes = one group per problem = the relaunch
    after a wave gave up its wait; the default (pipelined) run is what tests/test_gpu_parity.py pins to the goldens"""
    cases = [(f.split("/")[-1][:-5], spdg.load(f)) for f in S_FILES]
    cases = [(n, fx) for n, fx in cases if fx["prm"]["a_right"] - fx["prm"]["a_left"] >= 100]
    assert len(cases) >= 10
    key = lambda fx: (bool(fx["prm"]["local"]), fx["prm"]["max_vmf_space"], fx["prm"]["ubh"], fx["prm"]["sh"])
    n_batches = 0
    for kk in sorted({key(fx) for _, fx in cases}):
        sub = [(n, fx) for n, fx in cases if key(fx) == kk]
        ref = max((fx for _, fx in sub), key=lambda fx: fx["intpen"].size)
        sc = spdg.scoring(ref, scalar_engines=2, max_vmf_space=kk[1], ubh=kk[2], sh=kk[3])
        ps = abi.ProblemSet()
        for _, fx in sub:
            spdg.problem(fx, ps)

        def run():
            al = [(s, skl.ravel().tolist()) for s, skl in eng.align_s(sc, ps, allow_partial=True)]
            return al, [int(x) for x in eng.homscore_s(sc, ps)]
        monkeypatch.setenv("SPDP_A1_PIPE", "0")
        want = run()
        monkeypatch.delenv("SPDP_A1_PIPE")
        got = run()
        assert want == got, kk
        monkeypatch.setenv("SPDP_A0_PIPE_TEST_STALL", "1")
        again = run()
        monkeypatch.delenv("SPDP_A0_PIPE_TEST_STALL")
        assert want == again, kk
        n_batches += 1
    assert n_batches >= 3


def test_a1_linear_space_engine_against_oracle(eng, monkeypatch):
    """hirschbergS1 + its link walk on their own (test hook SPDP_UDH_ENGINE_A1 on spdp_scalar_udh), stripes pipelined and
    one group per problem, against the oracle's exact_udh: score, written-back ranges, cpos rows -- wherever the reference's
    walk is complete (a path down the window's left edge leaves rows of stale link lanes: DESIGN.md section 2)"""
    from oracle import oracle
    monkeypatch.setenv("SPDP_UDH_ENGINE_A1", "1")
    fx = spdg.load([f for f in S_FILES if f.endswith("s1_1400nt.spdg")][0])
    sc = spdg.scoring(fx, scalar_engines=2)
    n_cmp = n_skip = 0
    bad = []
    for n_im, m in ((1, 300), (4, 700), (7, 1100)):
        ps = _subranges(fx, 16, 320 + n_im, rows=(m, m))
        intvl = (m + n_im) // (n_im + 1)
        want = [oracle.exact_udh(sc, p, n_im) for p in ps.items]
        for pipe in ("0", "1"):
            monkeypatch.setenv("SPDP_A1_PIPE", pipe)
            scores, cpos, ranges, flags = eng.scalar_udh(sc, ps, n_im, intvl)
            for i, (ws, wc, wr) in enumerate(want):
                rows = [int(r[0]) for r in wc[:n_im]]
                if ws <= abi.NEVSEL or any(r == abi.END_OF_ULK for r in rows):
                    n_skip += 1
                    continue
                n_cmp += 1
                g = [[int(x) for x in r[:8]] for r in cpos[i][:n_im + 1]]
                w = [[int(x) for x in r[:8]] for r in wc[:n_im + 1]]
                if int(scores[i]) != ws or ranges[i].tolist() != [int(x) for x in wr] or g != w:
                    bad.append((n_im, pipe, i, int(scores[i]), ws, ranges[i].tolist(), [int(x) for x in wr], g[:2], w[:2]))
    assert not bad, bad[:3]
    assert n_cmp > 3 * n_skip and n_cmp >= 60
