"""Conserved intron positions (SpdpProblem.cip = Cip_score::cip_score(m), src/gsinfo.cc:65-79) pinned to REFERENCE runs:
the cp_* fixtures are `ref_dump -I` runs whose query carries a SigII (intron positions) with weight -J; the exact-model
engines (-A0: src/fwd2s1.cc:254, 338; -A1: src/fwd2s1_simd.cc:50) read the bonus, the `_wip` engines do not.  Oracle here,
GPU in tests/test_gpu_cip.py."""
import numpy as np
import pytest

from spaln_amd import abi
from tests import spdg
from tests.conftest import golden_files, golden_ids
from oracle import host_logic as hl


def cip_problem(fx, with_cip=True):
    ps = abi.ProblemSet()
    q = fx["prm"]
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    p = ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], q["a_left"], q["a_right"], q["b_left"], q["b_right"],
               (q["a_exgl"], q["a_exgr"], q["b_exgl"], q["b_exgr"]), cano5=fx["cano5"], cano3=fx["cano3"], dinc=dinc,
               cip=fx["cip"] if with_cip else None)
    p._owner = ps
    return ps, p


@pytest.mark.parametrize("path", golden_files("cp_"), ids=golden_ids("cp_"))
@pytest.mark.parametrize("alg,simd", [(0, 0), (1, 1), (2, 2)])
def test_cip_alignment_equals_reference(path, alg, simd):
    fx = spdg.load(path)
    sc = spdg.scoring(fx)
    _, p = cip_problem(fx)
    assert hl.homscore_s(sc, p, simd=simd) == int(fx[f"hom_scr_A{alg}"][0])
    scr, skl = hl.align_s(sc, p, simd=simd)
    assert scr == int(fx[f"aln_scr_A{alg}"][0])
    assert (skl or []) == fx[f"aln_skl_A{alg}"].tolist()


def test_the_bonus_matters_in_the_fixtures():
    """without the bonus row the exact-model alignment of at least two fixtures is a different one"""
    n = 0
    for path in golden_files("cp_"):
        fx = spdg.load(path)
        sc = spdg.scoring(fx)
        _, p = cip_problem(fx, with_cip=False)
        scr, skl = hl.align_s(sc, p, simd=0)
        n += scr != int(fx["aln_scr_A0"][0]) or (skl or []) != fx["aln_skl_A0"].tolist()
    assert n >= 2


@pytest.mark.parametrize("path", golden_files("cp_"), ids=[f.split("/")[-1][:-5] for f in golden_files("cp_")])
@pytest.mark.parametrize("alg", [0, 2])
def test_rescoring_carries_the_bonus(path, alg):
    """skl_rngS_ng with use_spb() (src/fwd2s1.cc:487, 615): total, statistics and exon records of the reference's run"""
    from oracle import host_logic
    fx = spdg.load(path)
    sc = spdg.scoring(fx)
    _, p = cip_problem(fx)
    fs = fx[f"rng_fstat_A{alg}"]
    h, fst, recs = host_logic.skl_rng_s(sc, p, [int(x) for x in fx[f"aln_skl_A{alg}"]], codonk1=fx["prm"]["codonk1"],
                                        minl=fx["prm"]["minl"], jneibr=int(fs[6]), lsg=int(fs[7]))
    assert h == int(fx[f"rng_scr_A{alg}"][0])
    assert fst == [int(x) for x in fs[:5]]
    assert recs == fx[f"rng_eij_A{alg}"].reshape(-1, 21).tolist()
