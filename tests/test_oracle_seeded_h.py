"""The protein seeded path (alignH_ng with algmode.qck = 1 .. 3: seededH_ng / interpolateH) against the reference, on the CPU.

The qh_* fixtures are `ref_dump -Q` runs of the compiled reference on aa x genome cases: HSPs of geneorient(), every
Wilip reply the reference's own walk received, score + SKL of alignH_ng under -A0 and -A2.  The walk under test is the
product's host source (spaln_amd/csrc/spdp_seeded_walk_h.h) compiled into oracle/libwalkcheck.so over the oracle's
protein ladder (lspH_ng, trcbkalignH_ng with / without introns and with a cut range)."""
import pytest

from spaln_amd import abi
from tests import spdg
from tests.conftest import golden_files
from oracle import seeded
from oracle import host_logic_h as hh

QH = golden_files("qh_")
# the reference's own -A2 traceback starts outside its bitmap on the tail of this case (an out-of-bounds read,
# SpdpAlignment n_skl = -2 in the product): its recorded output is not a function of the inputs
UNDEFINED = {("qh_0098", 2)}


def seeded_inputs_h(fx, alg):
    sc = spdg.scoring_h(fx)
    _, p = spdg.problem_h(fx)
    sp = abi.seed_params_from_fixture(fx)
    hsps, n = seeded.hsps_of(fx)
    return sc, sp, p, hsps, n, int(fx["seed_params"][1]), seeded.parse_wilip_log(fx[f"seed_wilip_A{alg}"])


@pytest.fixture(scope="module", params=QH, ids=[f.split("/")[-1][:-5] for f in QH])
def fx(request):
    f = spdg.load(request.param)
    f["_name"] = request.param.split("/")[-1][:-5]
    return f


@pytest.mark.parametrize("alg,simd", [(0, 0), (2, 2)])
def test_seeded_alignment_equals_reference(fx, alg, simd):
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs_h(fx, alg)
    if (fx["_name"], alg) in UNDEFINED:
        with pytest.raises(hh.ReferenceUndefined):
            seeded.align_h_seeded(sc, sp, p, hsps, n, lowest, wl, simd)
        return
    marks = {}
    scr, flat, rc = seeded.align_h_seeded(sc, sp, p, hsps, n, lowest, wl, simd, marks=marks)
    assert rc == 0
    assert scr == int(fx[f"seed_scr_A{alg}"][0])
    assert (flat or []) == fx[f"seed_skl_A{alg}"].tolist()
    # what the reference's walk left in its Exinon (phases at the junctions it chose itself; skl_rngH_ng reads them afterwards)
    want = {int(n): [int(a), int(b)] for n, a, b in fx[f"seed_marks_A{alg}"].reshape(-1, 3)}
    assert seeded.marks_changed(fx, marks) == want


def test_fixtures_reach_every_join():
    """every branch of interpolateH the walk serves is taken by some fixture (pick_unit -- bestwlu among several units
    -- needs paralogous HSP chains and is not reached by synthetic single-gene windows)"""
    joins = {}
    for f in QH:
        fx = spdg.load(f)
        sc, sp, p, hsps, n, lowest, wl = seeded_inputs_h(fx, 0)
        seeded.align_h_seeded(sc, sp, p, hsps, n, lowest, wl, 0, joins=joins)
    missing = [k for k in seeded.JOINS_H if not joins.get(k) and k not in ("pick_unit", "head_nogenome")]
    assert not missing, (missing, joins)


def test_dp_calls_are_made():
    """lspH_ng, trcbkalignH_ng with a cut range (shortcutH_ng) and without introns (the small-gap DP) all occur"""
    kinds = set()
    for f in QH:
        fx = spdg.load(f)
        sc, sp, p, hsps, n, lowest, wl = seeded_inputs_h(fx, 0)
        tr = []
        seeded.align_h_seeded(sc, sp, p, hsps, n, lowest, wl, 0, trace=tr)
        kinds |= {(kind, bool(a[11])) for kind, a, _ in tr}
    assert {(0, False), (1, True), (2, False), (3, False)} <= kinds


QH_A1 = golden_files("qh_a1_")


@pytest.mark.parametrize("path", QH_A1, ids=[f.split("/")[-1][:-5] for f in QH_A1])
def test_seeded_alignment_under_a1_equals_reference(path):
    """the protein walk with the -A1 engines (forwardH1 / hirschbergH1) behind its DP calls: `ref_dump -Q -A 0,1,2` runs"""
    fx = spdg.load(path)
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs_h(fx, 1)
    try:
        scr, flat, rc = seeded.align_h_seeded(sc, sp, p, hsps, n, lowest, wl, 1)
    except hh.ReferenceUndefined:
        pytest.skip("the reference's own -A1 traceback is undefined on a DP call of this case")
    assert rc == 0
    assert scr == int(fx["seed_scr_A1"][0])
    assert (flat or []) == fx["seed_skl_A1"].tolist()


LIVE_H = golden_files("live_h_")


@pytest.mark.parametrize("path", LIVE_H, ids=[f.split("/")[-1][:-5] for f in LIVE_H])
def test_live_pairs_equal_reference(path):
    """pairs taken out of whole-program runs (tools/dumpq_case.py: the reference's CLI wrote the fixture from inside its own
    alignH_ng call -- the window blkaln cut, the HSPs its block search left; only the program's -A0 run exists)"""
    fx = spdg.load(path)
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs_h(fx, 0)
    marks = {}
    scr, flat, rc = seeded.align_h_seeded(sc, sp, p, hsps, n, lowest, wl, 0, marks=marks)
    assert rc == 0 and scr == int(fx["seed_scr_A0"][0])
    assert (flat or []) == fx["seed_skl_A0"].tolist()
    want = {int(n): [int(a), int(b)] for n, a, b in fx["seed_marks_A0"].reshape(-1, 3)}
    assert want and seeded.marks_changed(fx, marks) == want


QHL3 = golden_files("qhl3_")


@pytest.mark.parametrize("path", QHL3, ids=[f.split("/")[-1][:-5] for f in QHL3])
def test_seeded_noll3_equals_reference(path):
    """the protein walk under double affine gaps (-yl3, -A0): forwardH_ng with and without a cut range and hirschbergH_ng
    with their F2 / E2 states behind it (round 5)"""
    fx = spdg.load(path)
    assert fx["prm"]["noll"] == 3
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs_h(fx, 0)
    marks = {}
    scr, flat, rc = seeded.align_h_seeded(sc, sp, p, hsps, n, lowest, wl, 0, marks=marks)
    assert rc == 0 and scr == int(fx["seed_scr_A0"][0])
    assert (flat or []) == fx["seed_skl_A0"].tolist()
    want = {int(n_): [int(a), int(b)] for n_, a, b in fx["seed_marks_A0"].reshape(-1, 3)}
    assert seeded.marks_changed(fx, marks) == want
