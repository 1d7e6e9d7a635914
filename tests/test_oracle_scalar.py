"""The oracle's scalar exact-ILD engines (oracle/spdp_oracle_scalar.c) against the reference's
-A0 output: HomScoreS_ng -> scorealoneS_ng on every fixture; alignS_ng -> forwardS_ng (score +
final SKL) on every fixture where the -A0 ladder takes the direct traceback."""
import pytest

from tests import spdg
from tests.conftest import golden_files, golden_ids
from oracle import oracle, host_logic


@pytest.fixture(scope="module", params=golden_files(), ids=golden_ids())
def fx(request):
    return spdg.load(request.param)


def direct_under_a0(fx, p, w):
    """lspS_ng with simd == 0 (src/fwd2s1.cc:1801-1838): does it call trcbkalignS_ng directly ?"""
    import numpy as np
    m, n = p.a_right - p.a_left, p.b_right - p.b_left
    if not m or not n or w.up == w.lw:
        return False
    if abs(n - m) < 8 or m == 1 or n == 1:
        return True
    f = np.float32
    k = f(w.lw - p.b_left + p.a_right)
    q = f(p.b_right - p.a_left - w.up)
    cvol = f(f(f(m) * f(n)) - f(f(k * k + q * q) / f(2)))
    return float(f(2.0) * cvol) < fx["prm"]["max_vmf_space"]


def test_scorealone(fx):
    sc = spdg.scoring(fx)
    ps, p = spdg.problem(fx)
    assert oracle.scalar_scorealone(sc, p) == int(fx["hom_scr_A0"][0])


def test_forward(fx):
    sc = spdg.scoring(fx)
    ps, p = spdg.problem(fx)
    w = oracle.stripe(p, sc.sh)
    if not direct_under_a0(fx, p, w):
        pytest.skip("-A0 ladder uses hirschbergS_ng here")
    scr, skl = oracle.scalar_forward(sc, p, w)
    rec = [(int(a), int(b)) for a, b in skl]
    fin = host_logic.trim_skl(host_logic.std_skl(rec), p) if len(rec) >= 2 else []
    flat = ([1, len(fin)] + [x for mn in fin for x in mn]) if fin else []
    assert scr == int(fx["aln_scr_A0"][0])
    assert flat == fx["aln_skl_A0"].tolist()


def test_align_a0_ladder(fx):
    """alignS_ng under -A0 end to end: lspS_ng with the hexagonal volume estimate, hirschbergS_ng with its
    recorded diagonal bounds as slab windows (mimd_postwork / rcsv_postwork), forwardS_ng per slab"""
    sc = spdg.scoring(fx)
    ps, p = spdg.problem(fx)
    scr, flat = host_logic.align_s(sc, p, simd=0)
    assert scr == int(fx["aln_scr_A0"][0])
    assert (flat or []) == fx["aln_skl_A0"].tolist()


def test_hirschberg_s_ng_is_exercised():
    """the fixtures above do reach the scalar linear-space engine, in recurrent and recursive mode"""
    seen = []
    orig = oracle.scalar_udh

    def spy(sc, p, n_im, intvl, w=None):
        seen.append(n_im)
        return orig(sc, p, n_im, intvl, w)
    oracle.scalar_udh = spy
    try:
        for name in ("s1_1400nt", "s1_forced_udh3"):
            fx = spdg.load([f for f in golden_files() if f.endswith(name + ".spdg")][0])
            sc = spdg.scoring(fx)
            _, p = spdg.problem(fx)
            host_logic.align_s(sc, p, simd=0)
    finally:
        oracle.scalar_udh = orig
    assert 7 in seen and seen.count(1) >= 3


# ---- double affine gaps (PwdB::Noll = 3, -yl3): forwardS_ng / scorealoneS_ng with their F2 / E2 states ------------------------
L3 = golden_files("l3_")


@pytest.mark.parametrize("path", L3, ids=golden_ids("l3_"))
def test_noll3_oracle_equals_reference(path):
    """HomScoreS_ng (scorealoneS_ng) and alignS_ng (forwardS_ng + traceback) of the reference under -yl3 -A0"""
    fx = spdg.load(path)
    assert fx["prm"]["noll"] == 3
    sc = spdg.scoring(fx, scalar_engines=1)
    _, p = spdg.problem(fx)
    assert oracle.scalar_scorealone(sc, p) == int(fx["hom_scr_A0"][0])
    scr, flat = host_logic.align_s(sc, p, simd=0)
    assert scr == int(fx["aln_scr_A0"][0]) and list(flat) == fx["aln_skl_A0"].tolist()
