"""The oracle's restatement of the alignS_ng dispatch ladder (oracle/host_logic.py)
against the reference's own alignS_ng output (-A2 and -A3 engine selectors):
raw engine score and the final SKL corner list, bit-identical."""
import os

import pytest

from tests import spdg
from tests.conftest import golden_files, golden_ids
from oracle import host_logic


@pytest.fixture(scope="module", params=golden_files(), ids=golden_ids())
def fx(request):
    return spdg.load(request.param)


@pytest.mark.parametrize("alg", [2, 3])
def test_align_s(fx, alg):
    sc = spdg.scoring(fx, nquant=(1 if alg == 3 else None))
    ps, p = spdg.problem(fx)
    try:
        scr, skl = host_logic.align_s(sc, p)
    except host_logic.NeedsScalarEngine:
        pytest.skip("needs the scalar engine (m < 8)")
    want = fx[f"aln_skl_A{alg}"].tolist()
    assert scr == int(fx[f"aln_scr_A{alg}"][0])
    assert (skl or []) == want



@pytest.mark.parametrize("alg", [0, 1, 2, 3])
def test_skl_rng_s_vs_reference(fx, alg):
    """skl_rngS_ng restated: total score, alignment statistics and per-exon records from the
    reference's own corner lists (every engine selector leaves the same kind of list)"""
    from oracle import host_logic
    if f"rng_eij_A{alg}" not in fx:
        pytest.skip("no alignment under this selector")
    sc = spdg.scoring(fx, nquant=(1 if alg == 3 else None))
    ps, p = spdg.problem(fx)
    fs = fx[f"rng_fstat_A{alg}"]
    h, fst, recs = host_logic.skl_rng_s(sc, p, [int(x) for x in fx[f"aln_skl_A{alg}"]],
                                        codonk1=fx["prm"]["codonk1"], minl=fx["prm"]["minl"],
                                        jneibr=int(fs[6]), lsg=int(fs[7]))
    assert h == int(fx[f"rng_scr_A{alg}"][0])
    assert fst == [int(x) for x in fs[:5]]
    assert recs == fx[f"rng_eij_A{alg}"].reshape(-1, 21).tolist()


@pytest.mark.parametrize("alg", [0, 1, 2, 3])
def test_homscore_s_ng_goldens(alg):
    """HomScoreS_ng as the reference returns it under -A0 / -A1 / -A2 / -A3, incl. the scalar branch below 4
    rows; -A1 is scoreonlyS1 (vector H / E / F with exact per-lane intron lists)"""
    from tests.conftest import golden_files
    n = 0
    for f in golden_files("s1_"):
        fx = spdg.load(f)
        sc = spdg.scoring(fx, nquant=1 if alg == 3 else None)
        _, p = spdg.problem(fx)
        assert host_logic.homscore_s(sc, p, simd=alg) == int(fx[f"hom_scr_A{alg}"][0]), f
        n += 1
    assert n >= 30


def test_align_a1_traceback_branch():
    """alignS_ng under -A1 end to end: forwardS1 (vector H / E / F, exact per-lane intron lists, Vmf pointers
    riding on the states) + Vmf::traceback in the traceback branch, hirschbergS1 + per-slab forwardS1 in the
    linear-space branches, stdskl / trimskl -- against the reference's -A1 output on every fixture"""
    from tests.conftest import golden_files
    n_ok = n_skip = 0
    for f in golden_files("s1_"):
        fx = spdg.load(f)
        sc = spdg.scoring(fx)
        _, p = spdg.problem(fx)
        try:
            scr, flat = host_logic.align_s(sc, p, simd=1)
        except host_logic.NeedsScalarEngine:
            n_skip += 1
            continue
        assert scr == int(fx["aln_scr_A1"][0]), f
        assert (flat or []) == fx["aln_skl_A1"].tolist(), f
        n_ok += 1
    assert n_ok == len(golden_files("s1_")) >= 34 and n_skip == 0


def test_align_a6_recursive_switch():
    """-A6 = the `_wip` engines with algmode.alg & 4: lspS_ng / lspH_ng always take the recursive
    linear-space branch once the traceback does not fit (SpdpScoring[H].recursive).  The harness runs it
    after -A3, whose engine constructor leaves IntronPrm.nquant = 1 behind (src/fwd2s1.cc:130), so the
    fixtures hold the flat-penalty model with the recursive ladder."""
    from tests.conftest import golden_files
    from oracle import host_logic_h
    for f in golden_files("s1_"):
        fx = spdg.load(f)
        sc = spdg.scoring(fx, nquant=1, recursive=1)
        _, p = spdg.problem(fx)
        scr, flat = host_logic.align_s(sc, p)
        assert scr == int(fx["aln_scr_A6"][0]) and (flat or []) == fx["aln_skl_A6"].tolist(), f
    for f in golden_files("h1_") + golden_files("c1_"):
        fx = spdg.load(f)
        if f.endswith("h1_cut_right.spdg") or f.endswith("h1_random.spdg"):
            continue                                   # undefined in the reference (see test_oracle_h_golden)
        sc = spdg.scoring_h(fx, nquant=1, recursive=1)
        _, p = spdg.problem_h(fx)
        scr, flat = host_logic_h.align_h(sc, p)
        assert scr == int(fx["aln_scr_A6"][0]) and (flat or []) == fx["aln_skl_A6"].tolist(), f


# ---- alignS_ng with its default orientation handling (ori = 3), pinned to the reference ----------------
O3 = golden_files("o3_")


@pytest.mark.parametrize("path", O3, ids=golden_ids("o3_"))
@pytest.mark.parametrize("alg", [2, 0])
def test_align_s_ori3_vs_reference(path, alg):
    """infer_orientation (src/fwd2s1.cc:2718-2730) + one alignment: which strand view wins, gsi->scr and the SKL
    (with A_RevCom in its header when the flipped pair was aligned), under -A2 and -A0; the reverse-strand problem is
    the reference's own comrev(a) / antiseq(b) with that strand's Exinon (ref_dump -O)"""
    fx = spdg.load(path)
    sc = spdg.scoring(fx)
    _, pf = spdg.problem(fx)
    _, pr = spdg.problem_rev(fx)
    (scr, skl), ori = host_logic.align_s_ori3(sc, pf, pr, simd=alg)
    assert ori == int(fx[f"ori3_rev_A{alg}"][0])
    assert scr == int(fx[f"ori3_scr_A{alg}"][0])
    assert (skl or []) == fx[f"ori3_skl_A{alg}"].tolist()


def test_ori3_fixtures_cover_both_outcomes():
    seen = {int(spdg.load(f)["ori3_rev_A2"][0]) for f in O3}
    assert seen == {0, 1} and len(O3) >= 5


def test_a1_double_affine_goldens():
    """-A1 with double affine gaps (-yl3, PwdB::Noll = 3; round 5): scoreonlyS1 and forwardS1 with their ev2 / fv2 vectors,
    five states a donor candidate can leave from and NCAND + 2 candidates per lane (src/fwd2s1_simd.cc:347-455, 556-755),
    against the compiled reference's own -yl3 -A1 runs (HomScoreS_ng, alignS_ng through the traceback branch of the ladder)."""
    from tests.conftest import golden_files
    n = 0
    for f in golden_files("l3a1_"):
        fx = spdg.load(f)
        sc = spdg.scoring(fx)
        assert sc.noll == 3
        _, p = spdg.problem(fx)
        assert host_logic.homscore_s(sc, p, simd=1) == int(fx["hom_scr_A1"][0]), f
        scr, flat = host_logic.align_s(sc, p, simd=1)
        assert scr == int(fx["aln_scr_A1"][0]), f
        assert (flat or []) == fx["aln_skl_A1"].tolist(), f
        n += 1
    assert n >= 5


def test_reference_hirschbergS1_under_yl3_is_not_usable():
    """Why the -A1 linear-space engine is refused with double affine gaps (spdp_api.cpp, DevRun::build): the compiled reference
    itself does not survive hirschbergS1 under -yl3 -- on the pairs of the l3a1_* fixtures a MaxVmfSpace small enough to send
    lspS_ng into the linear-space branch ends in heap corruption / SIGSEGV (glibc aborts: rc -6 / -11), where the
    traceback branch of the same pairs gives the fixtures' records.  Runs where the compiled reference is (oracle/_ref)."""
    import importlib.util
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref_dump = os.path.join(root, "oracle", "_ref", "ref_dump")
    tab = os.path.join(root, "oracle", "_ref", "table")
    if not (os.path.exists(ref_dump) and os.path.exists(os.path.join(tab, "mdm_mtx"))):
        pytest.skip("compiled reference not present")
    spec = importlib.util.spec_from_file_location("make_goldens", os.path.join(root, "tests", "golden", "make_goldens.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    from spaln_amd import synth
    cases = mg.cases()
    died = 0
    for name in ("l3a1_900nt", "l3a1_long_gaps"):
        w, q, opts = cases[name]
        opts = list(opts)
        opts[opts.index("-V") + 1] = "60000"
        with tempfile.TemporaryDirectory() as td:
            gf, qf, out = (os.path.join(td, x) for x in ("g.fa", "q.fa", "o.spdg"))
            synth.write_fasta(gf, "win", w)
            synth.write_fasta(qf, "qry", q)
            r = subprocess.run([ref_dump] + opts + [gf, qf, out], env=dict(os.environ, ALN_TAB=tab), capture_output=True)
        died += r.returncode in (-6, -11)
    assert died == 2
