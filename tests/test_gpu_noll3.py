"""Double affine gaps (Noll = 3, the reference's -yl3) in the -A0 engines: scorealoneS_ng and forwardS_ng as
spdp_rowwave<., ., ., DAGP> (five states per cell: H, E1, F1, E2, F2; src/fwd2s1.cc:48, 219-444, 1163-1336), hirschbergS_ng
as spdp_rowwave_udh<., DAGP> (a third plane of entries and of links per intermediate row; src/fwd2s1.cc:762-1104).
The reference's own HomScoreS_ng / alignS_ng under `-yl3 -A0` (tests/golden/l3_*.spdg; l3_udh_*: small MaxVmfSpace, the
ladder goes through the linear-space engine), then sub-ranges of the same pairs, ragged in height, one wave per problem
and as a pipeline of tiles, against the oracle."""
import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files, golden_ids
from spaln_amd import abi, synth

pytestmark = pytest.mark.gpu

L3_FILES = golden_files("l3_")


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("path", L3_FILES, ids=golden_ids("l3_"))
def test_noll3_equals_reference(eng, path):
    fx = spdg.load(path)
    assert fx["prm"]["noll"] == 3
    sc = spdg.scoring(fx, scalar_engines=1)
    ps, _ = spdg.problem(fx)
    assert int(eng.scalar_scorealone(sc, ps)[0]) == int(fx["hom_scr_A0"][0])
    assert int(eng.homscore_s(sc, ps)[0]) == int(fx["hom_scr_A0"][0])
    (scr, skl), = eng.align_s(sc, ps)
    assert scr == int(fx["aln_scr_A0"][0]) and skl.ravel().tolist() == fx["aln_skl_A0"].tolist()


def _subranges(fx, n, seed):
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + seed)
    extra = dict(cano5=fx["cano5"], cano3=fx["cano3"],
                 dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
    ps = abi.ProblemSet()
    for i in range(n):
        m = int(rng.integers(min(40, q["a_right"]), q["a_right"] + 1))
        al = int(rng.integers(0, q["a_right"] - m + 1))
        bl = int(rng.integers(0, max(1, min(400, q["b_right"] - m - 200))))
        br = int(rng.integers(max(bl + m + 100, q["b_right"] - 600), q["b_right"] + 1))
        exg = (1, 1, 1, 1) if i % 3 == 0 else tuple(int(x) for x in rng.integers(0, 2, size=4))
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, al + m, bl, br, exg, **extra)
    return ps


@pytest.mark.parametrize("name", ["l3_long_gaps", "l3_divergent", "l3_local"])
def test_noll3_subranges_against_oracle(eng, monkeypatch, name):
    from oracle import oracle
    f = [f for f in L3_FILES if f.endswith(name + ".spdg")]
    if not f:
        pytest.skip("fixture not present")
    fx = spdg.load(f[0])
    sc = spdg.scoring(fx, scalar_engines=1)
    ps = _subranges(fx, 12, 700 + len(name))
    want_f = [oracle.scalar_forward(sc, p) for p in ps.items]
    want_s = [oracle.scalar_scorealone(sc, p) for p in ps.items]
    for pipe in ("1", "0"):
        monkeypatch.setenv("SPDP_A0_PIPE", pipe)
        got = eng.scalar_forward(sc, ps)
        bad = [(i, s, ws, skl.tolist()[:8], wskl.tolist()[:8]) for i, ((s, skl), (ws, wskl)) in enumerate(zip(got, want_f))
               if s != ws or skl.tolist() != wskl.tolist()]
        assert not bad, (pipe, bad[:3])
        assert eng.scalar_scorealone(sc, ps).tolist() == want_s, pipe
    monkeypatch.delenv("SPDP_A0_PIPE")


@pytest.mark.parametrize("name,m,n_im", [("l3_udh_f2_cross", 529, 3), ("l3_udh_f2_cross", 529, 7), ("l3_udh_e2_on_row", 475, 3),
                                       ("l3_udh_900nt", 640, 5), ("l3_udh_local", 900, 4), ("l3_udh_divergent", 400, 2)])
def test_noll3_hirschberg_against_oracle(eng, monkeypatch, name, m, n_im):
    """hirschbergS_ng itself: scores, cpos rows, written-back ranges on sub-ranges of one height, against the oracle"""
    from oracle import oracle
    fx = spdg.load([f for f in L3_FILES if f.endswith(name + ".spdg")][0])
    q = fx["prm"]
    sc = spdg.scoring(fx, scalar_engines=1)
    rng = np.random.default_rng(synth.SEED + 900 + m + n_im)
    extra = dict(cano5=fx["cano5"], cano3=fx["cano3"],
                 dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
    ps = abi.ProblemSet()
    for i in range(10):
        al = int(rng.integers(0, q["a_right"] - m + 1))
        bl = int(rng.integers(0, 300))
        br = int(rng.integers(q["b_right"] - 400, q["b_right"] + 1))
        exg = (1, 1, 1, 1) if i % 3 == 0 else tuple(int(x) for x in rng.integers(0, 2, size=4))
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, al + m, bl, br, exg, **extra)
    intvl = (m + n_im) // (n_im + 1)
    want = [oracle.scalar_udh(sc, p, n_im, intvl) for p in ps.items]
    if name == "l3_udh_f2_cross":                        # the long insertion does cross an intermediate row in a gap state
        assert any(int(c[1]) == 1 for w in want if w[3] == 0 for c in w[1][:-1])
    for pipe in ("1", "0"):
        monkeypatch.setenv("SPDP_A0_PIPE", pipe)
        scores, cpos, ranges, flags = eng.scalar_udh(sc, ps, n_im, intvl)
        bad = []
        for i, (ws, wcpos, wrng, wflag) in enumerate(want):
            ok = int(flags[i]) == wflag
            if wflag == 0:
                ok = ok and int(scores[i]) == ws and ranges[i].tolist() == wrng.tolist() and cpos[i].tolist() == wcpos.tolist()
            if not ok:
                bad.append((i, int(scores[i]), ws, int(flags[i]), wflag))
        assert not bad, (pipe, bad[:3])
    monkeypatch.delenv("SPDP_A0_PIPE")


QL3 = golden_files("ql3_")


@pytest.mark.parametrize("path", QL3, ids=golden_ids("ql3_"))
def test_noll3_seeded_equals_reference(eng, path):
    """alignS_ng with seeding on (-Q5 .. -Q7) under -yl3: the walk's DP calls (lspS_ng through the ladder, trcbkalignS_ng
    with and without shortcutS_ng's cut range: spdp_rowwave<1, false, CUT, DAGP>) against the reference's own runs"""
    from tests.test_oracle_seeded import seeded_inputs
    fx = spdg.load(path)
    assert fx["prm"]["noll"] == 3
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs(fx, 0)
    sc.scalar_engines = 1
    res = eng.align_s_seeded(sc, sp, p._owner, [hsps if n else None], [lowest], [wl])
    scr, skl = res[0]
    assert scr == int(fx["seed_scr_A0"][0])
    assert ([int(x) for x in skl.ravel()] if len(skl) else []) == fx["seed_skl_A0"].tolist()
    st = eng.seeded_stats()
    assert st["walks"] == 1


def test_noll3_other_engines_refuse(eng):
    """the `_wip` engines (-A2 / -A3) are not built for Noll = 3: the upload says so (the -A1 pair is, since round 5:
    tests/test_gpu_noll3_a1.py; its linear-space form is refused there, as the reference's own crashes)"""
    fx = spdg.load([f for f in L3_FILES if f.endswith("l3_long_gaps.spdg")][0])
    ps, _ = spdg.problem(fx)
    with pytest.raises(Exception, match="Noll"):
        eng.homscore_s(spdg.scoring(fx, scalar_engines=0), ps)
