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
