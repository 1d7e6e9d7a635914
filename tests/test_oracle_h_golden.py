"""The aa x genome oracle (oracle/spdp_oracle_h.c, oracle/host_logic_h.py) against the reference's
own outputs (tests/golden/h1_*.spdg, written by oracle/ref_build/ref_dump_h.cc)."""
import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files
from oracle import oracle, host_logic_h as hh

H_FILES = golden_files("h1_") + golden_files("c1_")      # c1_: dictdisc proteins, species tables (BASELINE config 1)
# the reference starts its traceback outside its bitmap on this one (out-of-bounds read)
UNDEFINED = {"h1_cut_right", "h1_random"}


def _name(f):
    return f.split("/")[-1][:-5]


@pytest.mark.parametrize("path", H_FILES, ids=_name)
def test_stripe31(path):
    fx = spdg.load(path)
    sc = spdg.scoring_h(fx)
    _, p = spdg.problem_h(fx)
    w = oracle.stripe31(p, sc.sh)
    assert [w.lw, w.up, w.width] == [int(x) for x in fx["wdw"]]
    assert oracle.cells_h(p, w) > 0


@pytest.mark.parametrize("tag", ["qn", "q1"])
@pytest.mark.parametrize("path", H_FILES, ids=_name)
def test_forward_h1_wip(path, tag):
    fx = spdg.load(path)
    sc = spdg.scoring_h(fx, nquant=None if tag == "qn" else 1)
    _, p = spdg.problem_h(fx)
    s, skl, flag = oracle.wip_forward_h(sc, p)
    assert s == int(fx[f"wip_{tag}_fwd_scr"][0]) == int(fx[f"wip_{tag}_score"][0])
    assert skl.ravel().tolist() == fx[f"wip_{tag}_fwd_skl"].tolist()
    assert flag == (-3 if _name(path) in UNDEFINED else 0)


@pytest.mark.parametrize("alg", [2, 3])
@pytest.mark.parametrize("path", H_FILES, ids=_name)
def test_align_h(path, alg):
    fx = spdg.load(path)
    sc = spdg.scoring_h(fx, nquant=None if alg == 2 else 1)
    _, p = spdg.problem_h(fx)
    assert hh.homscore_h(sc, p) == int(fx[f"hom_scr_A{alg}"][0])
    if _name(path) in UNDEFINED:
        with pytest.raises(hh.ReferenceUndefined):
            hh.align_h(sc, p)
        return
    scr, skl = hh.align_h(sc, p)
    assert scr == int(fx[f"aln_scr_A{alg}"][0])
    assert (skl or []) == fx[f"aln_skl_A{alg}"].tolist()


def test_exon_structure_agrees_with_exact_engine():
    """on clean inputs the `_wip` model and the scalar exact model (-A0) find the same corners"""
    fx = spdg.load([f for f in H_FILES if f.endswith("h1_400aa.spdg")][0])
    assert fx["aln_skl_A2"].tolist() == fx["aln_skl_A0"].tolist()


def _udh_cases():
    import re
    out = []
    for f in H_FILES:
        fx = spdg.load(f)
        for k in fx:
            m = re.match(r"wip_(qn|q1)_udh(\d+)_scr", k)
            if m:
                out.append((f, m.group(1), int(m.group(2))))
    return out


@pytest.mark.parametrize("path,tag,n_im", _udh_cases(), ids=lambda v: _name(v) if isinstance(v, str) and "/" in v else str(v))
def test_hirschberg_h1_wip(path, tag, n_im):
    """score, cpos rows and written-back ranges of SimdAln2h1::hirschbergH1_wip"""
    fx = spdg.load(path)
    sc = spdg.scoring_h(fx, nquant=None if tag == "qn" else 1)
    _, p = spdg.problem_h(fx)
    s, cpos, rng = oracle.wip_udh_h(sc, p, n_im)
    assert s == int(fx[f"wip_{tag}_udh{n_im}_scr"][0])
    assert cpos.ravel().tolist() == fx[f"wip_{tag}_udh{n_im}_cpos"].tolist()
    assert rng.tolist() == fx[f"wip_{tag}_udh{n_im}_rng"][:4].tolist()



def _rescore_kw(fx):
    h = dict(zip(spdg.HPARAM_NAMES, (int(x) for x in fx["hparams"])))
    rp = [int(x) for x in fx["rparams"]]
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    return dict(intpen=fx["intpen"], t53=fx["t53"], dinc=dinc, lgop=fx["prm"]["lgop"], diffu=rp[0], k1=h["k1"],
                gape1=h["gape1"], gape2=h["gape2"], extragop=h["extragop"], minl=fx["prm"]["minl"],
                jneibr=rp[4], lcl=h["lcl"], lsg=rp[5], sup_tcodon=rp[1], many=rp[3])


@pytest.mark.parametrize("alg", [0, 2, 3])
@pytest.mark.parametrize("path", H_FILES, ids=_name)
def test_skl_rng_h_vs_reference(path, alg):
    """skl_rngH_ng restated: total score, statistics and per-exon / frame-shift records from the
    reference's own corner lists"""
    fx = spdg.load(path)
    if f"rng_eij_A{alg}" not in fx:
        pytest.skip("no alignment under this selector")
    if _name(path) == "h1_cut_right":
        pytest.skip("alignment lies beyond the window: the reference reads its heap (undefined)")
    if _name(path) == "h1_local_udh" and alg in (2, 3):
        pytest.skip("the -LS linear-space engine ends this alignment one row beyond the query (a reference quirk, "
                    "fwd2h1_wip_simd.h:652): the rescoring reads whatever lies behind the sequence")
    sc = spdg.scoring_h(fx, nquant=None if alg != 3 else 1)
    _, p = spdg.problem_h(fx)
    h, fst, recs = hh.skl_rng_h(sc, p, [int(x) for x in fx[f"aln_skl_A{alg}"]], **_rescore_kw(fx))
    assert h == int(fx[f"rng_scr_A{alg}"][0])
    assert fst == [int(x) for x in fx[f"rng_fstat_A{alg}"][:5]]
    assert recs == fx[f"rng_eij_A{alg}"].reshape(-1, 21).tolist()


@pytest.mark.parametrize("path", H_FILES, ids=_name)
def test_align_h_a1(path):
    """-A1: alignH_ng over forwardH1 / hirschbergH1 (full-precision intron lengths, 16-bit lanes) against the
    reference's own -A1 run (differs from both -A0 and -A2 on a third of the fixtures)"""
    fx = spdg.load(path)
    if "aln_skl_A1" not in fx:
        pytest.skip("no -A1 record")
    sc = spdg.scoring_h(fx)
    _, p = spdg.problem_h(fx)
    scr, skl = hh.align_h(sc, p, simd=1)
    assert scr == int(fx["aln_scr_A1"][0])
    assert (skl or []) == fx["aln_skl_A1"].tolist()
