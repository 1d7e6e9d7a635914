"""The scalar aa x genome engine of the oracle (oracle/spdp_oracle_h_scalar.c: forwardH_ng + initH_ng /
lastH_ng + Vmf traceback) against the reference's -A0 outputs in tests/golden/h1_*.spdg, and the
below-8-rows fallback of the -A2 / -A3 dispatch."""
import pytest

from tests import spdg
from tests.conftest import golden_files
from oracle import oracle, host_logic_h as hh

H_FILES = golden_files("h1_") + golden_files("c1_")      # c1_: dictdisc proteins, species tables (BASELINE config 1)


def _name(f):
    return f.split("/")[-1][:-5]


@pytest.mark.parametrize("path", H_FILES, ids=_name)
def test_homscore_a0(path):
    """HomScoreH_ng under -A0 = forwardH_ng without a Vmf (src/fwd2h1.cc:3297)"""
    fx = spdg.load(path)
    sc = spdg.scoring_h(fx)
    _, p = spdg.problem_h(fx)
    assert hh.homscore_h(sc, p, simd=0) == int(fx["hom_scr_A0"][0])


@pytest.mark.parametrize("path", H_FILES, ids=_name)
def test_align_a0(path):
    """alignH_ng under -A0 end to end: forwardH_ng + Vmf::traceback in the traceback branch, hirschbergH_ng
    with its recorded diagonal bounds as slab windows in the linear-space branch, stdskl3"""
    fx = spdg.load(path)
    sc = spdg.scoring_h(fx)
    _, p = spdg.problem_h(fx)
    scr, flat = hh.align_h(sc, p, simd=0)
    assert scr == int(fx["aln_scr_A0"][0])
    assert (flat or []) == fx["aln_skl_A0"].tolist()


@pytest.mark.parametrize("alg", [2, 3])
@pytest.mark.parametrize("m", [3, 5, 7])
def test_below_8_rows_default_modes(m, alg):
    """-A2 / -A3 run the scalar engine below 8 query rows (src/fwd2h1.cc:2005, 3297)"""
    path = [f for f in H_FILES if _name(f) == f"h1_tiny_m{m}"][0]
    fx = spdg.load(path)
    sc = spdg.scoring_h(fx, nquant=1 if alg == 3 else None)
    _, p = spdg.problem_h(fx)
    assert hh.homscore_h(sc, p, simd=alg) == int(fx[f"hom_scr_A{alg}"][0])
    scr, flat = hh.align_h(sc, p, simd=alg)
    assert scr == int(fx[f"aln_scr_A{alg}"][0])
    assert flat == fx[f"aln_skl_A{alg}"].tolist()


def test_hirschberg_h_ng_is_exercised():
    seen = []
    orig = oracle.scalar_udh_h

    def spy(sc, p, n_im, intvl, w=None):
        seen.append(n_im)
        return orig(sc, p, n_im, intvl, w)
    oracle.scalar_udh_h = spy
    try:
        for name in ("h1_450aa_auto", "h1_forced_udh3", "h1_auto_udh"):
            fx = spdg.load([f for f in H_FILES if _name(f) == name][0])
            sc = spdg.scoring_h(fx)
            _, p = spdg.problem_h(fx)
            hh.align_h(sc, p, simd=0)
    finally:
        oracle.scalar_udh_h = orig
    assert len(seen) >= 3 and max(seen) >= 2


# ---- double affine gaps (PwdB::Noll = 3, -yl3): forwardH_ng / hirschbergH_ng with their F2 / E2 states (round 5) --------------
HL3 = golden_files("hl3_")


@pytest.mark.parametrize("path", HL3, ids=_name)
def test_noll3_equals_reference(path):
    """the reference's own HomScoreH_ng / alignH_ng under `-yl3 -A0` (hl3_udh_*: a small MaxVmfSpace sends the ladder
    through hirschbergH_ng with its three planes of links per intermediate row)"""
    fx = spdg.load(path)
    assert fx["prm"]["noll"] == 3
    sc = spdg.scoring_h(fx)
    _, p = spdg.problem_h(fx)
    assert hh.homscore_h(sc, p, simd=0) == int(fx["hom_scr_A0"][0])
    scr, flat = hh.align_h(sc, p, simd=0)
    assert scr == int(fx["aln_scr_A0"][0])
    assert (flat or []) == fx["aln_skl_A0"].tolist()


def test_noll3_long_states_decide():
    """the fixtures are not decided by the affine pair alone: with Noll = 2 most of them score lower, and the linear-space
    engine is reached with three planes"""
    lower = 0
    for path in HL3:
        fx = spdg.load(path)
        _, p = spdg.problem_h(fx)
        s2, _ = hh.align_h(spdg.scoring_h(fx, noll=2), p, simd=0)
        lower += s2 < int(fx["aln_scr_A0"][0])
    assert lower >= 8
    seen = []
    orig = oracle.scalar_udh_h

    def spy(sc, p, n_im, intvl, w=None):
        seen.append((sc.noll, n_im))
        return orig(sc, p, n_im, intvl, w)
    oracle.scalar_udh_h = spy
    try:
        for name in ("hl3_udh_auto", "hl3_udh_forced7", "hl3_udh_450aa", "hl3_udh_local"):
            fx = spdg.load([f for f in HL3 if _name(f) == name][0])
            _, p = spdg.problem_h(fx)
            hh.align_h(spdg.scoring_h(fx), p, simd=0)
    finally:
        oracle.scalar_udh_h = orig
    assert len(seen) >= 4 and all(n == 3 for n, _ in seen) and max(k for _, k in seen) >= 3
