"""The seeded path (alignS_ng with algmode.qck = 1 .. 3, -Q5 .. -Q7) against the reference, on the CPU.

The q_* fixtures are `ref_dump -Q` runs of the compiled reference: the HSPs geneorient() found, every Wilip reply
its own seeded walk received at the recursion levels, and score + SKL of alignS_ng under -A0 and -A2.  The walk under
test is the product's host source (spaln_amd/csrc/spdp_seeded_walk.h) compiled into oracle/libwalkcheck.so with the
oracle's DP ladder where the product has the device; tests/test_gpu_seeded.py runs the same source with the GPU behind it."""
import numpy as np
import pytest

from spaln_amd import abi
from tests import spdg
from tests.conftest import golden_files, golden_ids
from oracle import seeded


def seeded_inputs(fx, alg):
    sc = spdg.scoring(fx)
    ps = abi.ProblemSet()
    _, p = spdg.problem(fx, ps)
    h5, h3 = np.ascontiguousarray(fx["phs5"]), np.ascontiguousarray(fx["phs3"])
    p.phs5, p.phs3 = h5.ctypes.data, h3.ctypes.data
    p._phs = (h5, h3)
    sp = abi.seed_params_from_fixture(fx)
    hsps, n = seeded.hsps_of(fx)
    return sc, sp, p, hsps, n, int(fx["seed_params"][1]), seeded.parse_wilip_log(fx[f"seed_wilip_A{alg}"])


Q_FILES = [f for f in golden_files("q_") if "/q_o3_" not in f]         # (q_o3_*: ori = 3 runs, tested on their own below)


@pytest.fixture(scope="module", params=Q_FILES, ids=[f.split("/")[-1][:-5] for f in Q_FILES])
def fx(request):
    return spdg.load(request.param)


@pytest.mark.parametrize("alg,simd", [(0, 0), (2, 2)])
def test_seeded_alignment_equals_reference(fx, alg, simd):
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs(fx, alg)
    scr, flat, rc = seeded.align_s_seeded(sc, sp, p, hsps, n, lowest, wl, simd)
    assert rc == 0
    assert scr == int(fx[f"seed_scr_A{alg}"][0])
    assert (flat or []) == fx[f"seed_skl_A{alg}"].tolist()


QL3 = golden_files("ql3_")


@pytest.mark.parametrize("path", QL3, ids=golden_ids("ql3_"))
def test_seeded_noll3_equals_reference(path):
    """the seeded path under double affine gaps (-yl3; -A0 engines behind the walk): GapPenalty's switch to the long pair
    beyond codonk1 in the closed-form joins, E2 / F2 in every DP call"""
    fx = spdg.load(path)
    assert fx["prm"]["noll"] == 3
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs(fx, 0)
    scr, flat, rc = seeded.align_s_seeded(sc, sp, p, hsps, n, lowest, wl, 0)
    assert rc == 0 and scr == int(fx["seed_scr_A0"][0])
    assert (flat or []) == fx["seed_skl_A0"].tolist()


def test_fixtures_reach_every_join():
    """every branch of interpolateS (and bestwlu) is taken by some fixture, most by several"""
    joins = {}
    for f in Q_FILES:
        fx = spdg.load(f)
        sc, sp, p, hsps, n, lowest, wl = seeded_inputs(fx, 2)
        seeded.align_s_seeded(sc, sp, p, hsps, n, lowest, wl, 2, joins=joins)
    missing = [k for k in seeded.JOINS if not joins.get(k)]
    assert not missing, (missing, joins)


def test_dp_calls_are_made():
    """the walk does hand gaps between HSPs to the DP engines (lspS_ng, trcbkalignS_ng with and without a cut range)"""
    kinds = set()
    for name in ("q_0687", "q_0745", "q_c2_seed0"):
        fx = spdg.load([f for f in golden_files("q_") if f.endswith(name + ".spdg")][0])
        sc, sp, p, hsps, n, lowest, wl = seeded_inputs(fx, 2)
        tr = []
        seeded.align_s_seeded(sc, sp, p, hsps, n, lowest, wl, 2, trace=tr)
        for kind, a, _ in tr:
            kinds.add((kind, bool(a[11])))
    assert {(0, False), (1, False), (1, True), (2, False)} <= kinds


O3 = [f for f in golden_files("q_o3_")]


@pytest.mark.parametrize("path", O3, ids=[f.split("/")[-1][:-5] for f in O3])
@pytest.mark.parametrize("alg,simd", [(0, 0), (2, 2)])
def test_seeded_ori3_equals_reference(path, alg, simd):
    """alignS_ng(seqs, pwd, gsi, 3) with seeding on: both strands walked (the reverse one with the HSP list turned around),
    the reverse kept only if strictly better, A_RevCom in its header"""
    fx = spdg.load(path)
    sc, sp, p, hsps, n, lowest, wl_f = seeded_inputs(fx, alg)
    _, pr = spdg.problem_rev(fx)
    h5, h3 = np.ascontiguousarray(fx["r_phs5"]), np.ascontiguousarray(fx["r_phs3"])
    pr.phs5, pr.phs3 = h5.ctypes.data, h3.ctypes.data
    wl_r = seeded.parse_wilip_log(fx[f"seed_wilip_A{alg}"], strand=1)
    scr, flat, rev = seeded.align_s_seeded_ori3(sc, sp, p, pr, hsps, n, lowest, wl_f, wl_r, simd)
    assert rev == int(fx[f"seed_rev_A{alg}"][0])
    assert scr == int(fx[f"seed_scr_A{alg}"][0])
    assert (flat or []) == fx[f"seed_skl_A{alg}"].tolist()


def test_ori3_fixtures_take_both_orientations():
    assert sorted({int(spdg.load(f)["seed_rev_A2"][0]) for f in O3}) == [0, 1]


def test_annotated_intron_positions_enter_the_junction_rule():
    """use_spb(): q_cip0 carries the query's intron positions at the true junctions (ref_dump -I / -J); indelfreespjS adds
    their bonus (src/fwd2s1.cc:2030-2037) -- without it the walk falls 5 x 100 short of the reference's score"""
    fx = spdg.load([f for f in golden_files("q_cip") if f.endswith("q_cip0.spdg")][0])
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs(fx, 2)
    assert seeded.align_s_seeded(sc, sp, p, hsps, n, lowest, wl, 2)[0] == int(fx["seed_scr_A2"][0])
    p.cip = None
    assert seeded.align_s_seeded(sc, sp, p, hsps, n, lowest, wl, 2)[0] == int(fx["seed_scr_A2"][0]) - 500


A1 = [f for f in golden_files("q_a1_")]


@pytest.mark.parametrize("path", A1, ids=[f.split("/")[-1][:-5] for f in A1])
def test_seeded_alignment_under_a1_equals_reference(path):
    """the same walk with the -A1 engines (forwardS1 / hirschbergS1) behind its DP calls: `ref_dump -Q -A 0,1,2` runs"""
    fx = spdg.load(path)
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs(fx, 1)
    scr, flat, rc = seeded.align_s_seeded(sc, sp, p, hsps, n, lowest, wl, 1)
    assert rc == 0
    assert scr == int(fx["seed_scr_A1"][0])
    assert (flat or []) == fx["seed_skl_A1"].tolist()
