"""GPU parity of the scalar aa x genome engine (spdp_scalar_forward_h = forwardH_ng + trcbkalignH_ng's
record hand-over, spdp_h_scalar.hip) and of the below-8-rows branch of the dispatch, against the
reference's -A0 goldens and the oracle."""
import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi, synth

pytestmark = pytest.mark.gpu

H_FILES = golden_files("h1_")


def _name(f):
    return f.split("/")[-1][:-5]


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


def _batch(local):
    cases = [(_name(f), spdg.load(f)) for f in H_FILES if (_name(f) == "h1_local") == local]
    sc = spdg.scoring_h(max((fx for _, fx in cases), key=lambda fx: fx["intpen"].size))
    ps = abi.ProblemSetH()
    for _, fx in cases:
        spdg.problem_h(fx, ps)
    return cases, sc, ps


@pytest.mark.parametrize("local", [False, True])
def test_scores_against_reference_a0(eng, local):
    """HomScoreH_ng under -A0 = forwardH_ng without a Vmf: every fixture, one batch"""
    cases, sc, ps = _batch(local)
    res = eng.scalar_forward_h(sc, ps, traceback=False)
    bad = [(name, s, int(fx["hom_scr_A0"][0])) for (name, fx), (s, _) in zip(cases, res)
           if s != int(fx["hom_scr_A0"][0])]
    assert not bad, bad


@pytest.mark.parametrize("local", [False, True])
def test_records_against_oracle(eng, local):
    """raw Mfile records of trcbkalignH_ng's scalar branch (the oracle is pinned on the -A0 alignments)"""
    from oracle import oracle
    cases, sc, ps = _batch(local)
    res = eng.scalar_forward_h(sc, ps)
    bad = []
    for (name, fx), p, (s, skl) in zip(cases, ps.items, res):
        ws, wskl = oracle.scalar_forward_h(sc, p)
        if s != ws or skl.tolist() != wskl.tolist():
            bad.append((name, s, ws, skl.ravel().tolist()[:12], wskl.ravel().tolist()[:12]))
    assert not bad, bad[:4]


@pytest.mark.parametrize("alg", [2, 3])
def test_below_8_rows_goldens(eng, alg):
    """-A2 / -A3 on queries below 8 residues: alignH_ng and HomScoreH_ng go through the scalar engine"""
    cases = [(m, spdg.load([f for f in H_FILES if _name(f) == f"h1_tiny_m{m}"][0])) for m in (3, 5, 7)]
    sc = spdg.scoring_h(cases[0][1], nquant=None if alg == 2 else 1)
    ps = abi.ProblemSetH()
    for _, fx in cases:
        spdg.problem_h(fx, ps)
    res = eng.align_h(sc, ps)
    hom = eng.homscore_h(sc, ps)
    for (m, fx), (score, skl, flag), hs in zip(cases, res, hom):
        assert flag == 0 and score == int(fx[f"aln_scr_A{alg}"][0]), m
        assert skl.ravel().tolist() == fx[f"aln_skl_A{alg}"].tolist(), m
        assert int(hs) == int(fx[f"hom_scr_A{alg}"][0]), m


def test_small_subranges_against_oracle(eng):
    """random sub-ranges with 1 .. 7 query rows and all end-gap flag combinations: engine and dispatch"""
    from oracle import oracle, host_logic_h as hh
    fx = spdg.load([f for f in H_FILES if f.endswith("h1_400aa.spdg")][0])
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 81)
    sc = spdg.scoring_h(fx)
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    ps = abi.ProblemSetH()
    for i in range(160):
        m = int(rng.integers(1, 8))
        al = int(rng.integers(0, q["a_right"] - m))
        bl = int(rng.integers(1, q["b_right"] - 700))
        br = bl + int(rng.integers(max(3 * m + 2, 12), 600))
        exg = tuple(int(x) for x in rng.integers(0, 3 if i % 4 == 0 else 2, size=4))
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
               fx["phs5"], fx["phs3"], al, al + m, bl, br, exg, exin=(q["b_left"], q["b_right"]), dinc=dinc)
    res = eng.scalar_forward_h(sc, ps)
    bad = []
    for i, (p, (s, skl)) in enumerate(zip(ps.items, res)):
        ws, wskl = oracle.scalar_forward_h(sc, p)
        if s != ws or skl.tolist() != wskl.tolist():
            bad.append((i, (p.a_left, p.a_right, p.b_left, p.b_right), (p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr),
                        s, ws, skl.ravel().tolist()[:10], wskl.ravel().tolist()[:10]))
    assert not bad, bad[:4]
    # the same problems through alignH_ng
    res = eng.align_h(sc, ps)
    n_ok = 0
    for i, (p, (score, skl, flag)) in enumerate(zip(ps.items, res)):
        try:
            ws, wskl = hh.align_h(sc, p)
        except hh.NotRestated:
            assert flag == 1, i
            continue
        n_ok += 1
        if flag != 0 or score != ws or skl.ravel().tolist() != (wskl or []):
            bad.append((i, (p.a_left, p.a_right, p.b_left, p.b_right), flag, score, ws,
                        skl.ravel().tolist()[:12], (wskl or [])[:12]))
    assert n_ok >= 100 and not bad, bad[:4]
