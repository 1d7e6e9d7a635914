"""Double affine gaps (Noll = 3, the reference's -yl3) in the protein -A0 engines (round 5): forwardH_ng as
spdh_rowwave<0 / 1, ., CUT, DAGP> and hirschbergH_ng as spdh_rowwave<2, ., ., DAGP> (a second deletion state by diagonal, a
second insertion queue, five states a donor candidate leaves from, three planes of links per intermediate row;
src/fwd2h1.cc:297-617, 1088-1520).  The reference's own HomScoreH_ng / alignH_ng under `-yl3 -A0` (tests/golden/hl3_*;
hl3_udh_*: small MaxVmfSpace, the ladder goes through the linear-space engine), its seeded runs (qhl3_*: the walk's DP
calls incl. the cut range), then sub-ranges against the oracle."""
import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi, synth

pytestmark = pytest.mark.gpu

HL3 = golden_files("hl3_")
QHL3 = golden_files("qhl3_")


def _name(f):
    return f.split("/")[-1][:-5]


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("path", HL3, ids=_name)
def test_noll3_equals_reference(eng, path):
    fx = spdg.load(path)
    assert fx["prm"]["noll"] == 3
    sc = spdg.scoring_h(fx, scalar_engines=1)
    ps, _ = spdg.problem_h(fx)
    assert int(eng.homscore_h(sc, ps)[0]) == int(fx["hom_scr_A0"][0])
    (scr, skl, flag), = eng.align_h(sc, ps)
    assert flag == 0 and scr == int(fx["aln_scr_A0"][0])
    assert skl.ravel().tolist() == fx["aln_skl_A0"].tolist()


def test_noll3_batch(eng):
    """the fixtures of one parameter set in one call (two problems per block, ragged sizes)"""
    for local in (False, True):
        cases = [spdg.load(f) for f in HL3 if bool(spdg.load(f)["prm"]["local"]) == local
                 and spdg.load(f)["prm"]["max_vmf_space"] == spdg.load(HL3[0])["prm"]["max_vmf_space"]]
        if not cases:
            continue
        sc = spdg.scoring_h(max(cases, key=lambda fx: fx["intpen"].size), scalar_engines=1)
        ps = abi.ProblemSetH()
        for fx in cases:
            spdg.problem_h(fx, ps)
        res = eng.align_h(sc, ps)
        hom = eng.homscore_h(sc, ps)
        for fx, (scr, skl, flag), hs in zip(cases, res, hom):
            assert flag == 0 and scr == int(fx["aln_scr_A0"][0]) and skl.ravel().tolist() == fx["aln_skl_A0"].tolist()
            assert int(hs) == int(fx["hom_scr_A0"][0])


def _subranges(fx, n, seed, m_lo=40):
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + seed)
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    ps = abi.ProblemSetH()
    for i in range(n):
        m = int(rng.integers(min(m_lo, q["a_right"]), q["a_right"] + 1))
        al = int(rng.integers(0, q["a_right"] - m + 1))
        bl = int(rng.integers(1, max(2, min(400, q["b_right"] - 3 * m - 300))))
        br = int(rng.integers(max(bl + 3 * m + 100, q["b_right"] - 600), q["b_right"] + 1))
        exg = (1, 1, 1, 1) if i % 3 == 0 else tuple(int(x) for x in rng.integers(0, 2, size=4))
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
               fx["phs5"], fx["phs3"], al, al + m, bl, br, exg, exin=(q["b_left"], q["b_right"]), dinc=dinc)
    return ps


@pytest.mark.parametrize("name", ["hl3_long_gaps", "hl3_divergent", "hl3_local", "hl3_window_deletions"])
def test_noll3_subranges_against_oracle(eng, name):
    """forwardH_ng on sub-ranges, ragged in height (one to five tiles of rows), all end-gap flag combinations: scores without
    a Vmf and the raw records with one"""
    from oracle import oracle
    fx = spdg.load([f for f in HL3 if _name(f) == name][0])
    sc = spdg.scoring_h(fx, scalar_engines=1)
    ps = _subranges(fx, 14, 1700 + len(name))
    got = eng.scalar_forward_h(sc, ps)
    got_s = eng.scalar_forward_h(sc, ps, traceback=False)
    bad = []
    for i, (p, (s, skl), (s0, _)) in enumerate(zip(ps.items, got, got_s)):
        ws, wskl = oracle.scalar_forward_h(sc, p)
        if s != ws or s0 != ws or skl.tolist() != wskl.tolist():
            bad.append((i, (p.a_left, p.a_right, p.b_left, p.b_right), s, s0, ws, skl.ravel().tolist()[:10], wskl.ravel().tolist()[:10]))
    assert not bad, bad[:3]


@pytest.mark.parametrize("name,m,n_im", [("hl3_udh_auto", 200, 3), ("hl3_udh_auto", 260, 7), ("hl3_udh_450aa", 300, 2),
                                       ("hl3_udh_local", 180, 4), ("hl3_long_gaps", 150, 1)])
def test_noll3_hirschberg_against_oracle(eng, name, m, n_im):
    """hirschbergH_ng itself: scores, cpos rows, written-back ranges on sub-ranges of one height, against the oracle"""
    from oracle import oracle
    fx = spdg.load([f for f in HL3 if _name(f) == name][0])
    q = fx["prm"]
    sc = spdg.scoring_h(fx, scalar_engines=1)
    rng = np.random.default_rng(synth.SEED + 1900 + m + n_im)
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    ps = abi.ProblemSetH()
    for i in range(10):
        al = int(rng.integers(0, q["a_right"] - m + 1))
        bl = int(rng.integers(1, 300))
        br = int(rng.integers(q["b_right"] - 400, q["b_right"] + 1))
        exg = (1, 1, 1, 1) if i % 3 == 0 else tuple(int(x) for x in rng.integers(0, 2, size=4))
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
               fx["phs5"], fx["phs3"], al, al + m, bl, br, exg, exin=(q["b_left"], q["b_right"]), dinc=dinc)
    intvl = (m + n_im) // (n_im + 1)
    want = [oracle.scalar_udh_h(sc, p, n_im, intvl) for p in ps.items]
    scores, cpos, ranges, flags = eng.scalar_udh_h(sc, ps, n_im, intvl)
    bad = []
    for i, (ws, wcpos, wrng, wflag) in enumerate(want):
        ok = int(flags[i]) == wflag
        if wflag == 0:
            ok = ok and int(scores[i]) == ws and ranges[i].tolist() == wrng.tolist() and cpos[i].tolist() == wcpos.tolist()
        if not ok:
            bad.append((i, int(scores[i]), ws, int(flags[i]), wflag, ranges[i].tolist(), wrng.tolist()))
    assert not bad, bad[:3]
    assert sum(1 for w in want if w[3] == 0) >= 5


@pytest.mark.parametrize("path", QHL3, ids=_name)
def test_noll3_seeded_equals_reference(eng, path):
    """alignH_ng with seeding on under -yl3: the walk's DP calls (lspH_ng through the ladder, trcbkalignH_ng with and without
    shortcutH_ng's cut range: spdh_rowwave<1, false, CUT, DAGP>) against the reference's own runs"""
    from oracle import seeded
    from tests.test_oracle_seeded_h import seeded_inputs_h
    fx = spdg.load(path)
    assert fx["prm"]["noll"] == 3
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs_h(fx, 0)
    sc.scalar_engines = 1
    (scr, skl), = eng.align_h_seeded(sc, sp, p._owner, [hsps if n else None], [lowest], [wl])
    assert scr == int(fx["seed_scr_A0"][0])
    assert ([int(x) for x in skl.ravel()] if len(skl) else []) == fx["seed_skl_A0"].tolist()
    want = {int(n_): [int(a), int(b)] for n_, a, b in fx["seed_marks_A0"].reshape(-1, 3)}
    assert seeded.marks_changed(fx, eng.seeded_phase_marks(0)) == want


def test_noll3_other_engines_refuse(eng):
    """the `_wip` and -A1 protein engines are not built for Noll = 3: the upload says so"""
    fx = spdg.load([f for f in HL3 if _name(f) == "hl3_long_gaps"][0])
    ps, _ = spdg.problem_h(fx)
    for se in (0, 2):
        with pytest.raises(Exception, match="noll|Noll"):
            eng.homscore_h(spdg.scoring_h(fx, scalar_engines=se), ps)
