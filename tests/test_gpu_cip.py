"""SpdpProblem.cip (Cip_score::cip_score(m), src/gsinfo.h:128-140): the per-row bonus every intron accepted in that row
earns under the -A0 engines (src/fwd2s1.cc:254, 338) and the -A1 engines (src/fwd2s1_simd.cc:50).  GPU against the
oracle's restatement with random sparse bonuses; the bonus must matter (alignments change), and a NULL list must
equal an all-zero one.  Parity with the reference itself: test_cip_pinned_to_reference_runs (the cp_* fixtures, `ref_dump -I`:
the query carries a SigII as a database entry with gene structure would)."""
import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files

pytestmark = pytest.mark.gpu


def _cases(rng, n_sub):
    """sub-ranges of one fixture, each with its own bonus row"""
    from spaln_amd import abi
    fx = spdg.load([f for f in golden_files("s1_") if f.endswith("s1_1400nt.spdg")][0])
    q = fx["prm"]
    extra = dict(cano5=fx["cano5"], cano3=fx["cano3"],
                 dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
    plain, bonus, zero = abi.ProblemSet(), abi.ProblemSet(), abi.ProblemSet()
    for i in range(n_sub):
        al = int(rng.integers(0, 300))
        ar = int(rng.integers(al + 400, min(al + 900, q["a_right"]) + 1))
        bl = int(rng.integers(0, 400))
        br = int(rng.integers(q["b_right"] - 1000, q["b_right"] + 1))
        cip = np.zeros(fx["a_codes"].size + 1, dtype=np.int32)
        hit = rng.random(cip.size) < 0.06
        cip[hit] = rng.integers(40, 400, size=int(hit.sum()))
        args = (fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, ar, bl, br, (1, 1, 1, 1))
        plain.add(*args, **extra)
        bonus.add(*args, **extra, cip=cip)
        zero.add(*args, **extra, cip=np.zeros_like(cip))
    return fx, plain, bonus, zero


@pytest.mark.parametrize("engines", [1, 2])
def test_cip_against_oracle(engines):
    from spaln_amd import engine
    from oracle import host_logic as hl
    rng = np.random.default_rng(77 + engines)
    fx, plain, bonus, zero = _cases(rng, 10)
    eng = engine.Engine(0)
    changed = 0
    for vmf in (60000, 33554432):                           # linear-space ladder, and one traceback per problem
        sc = spdg.scoring(fx, scalar_engines=engines, max_vmf_space=vmf)
        r_plain = eng.align_s(sc, plain, allow_partial=True)
        r_bonus = eng.align_s(sc, bonus, allow_partial=True)
        r_zero = eng.align_s(sc, zero, allow_partial=True)
        h_bonus = eng.homscore_s(sc, bonus, allow_partial=True)
        for i, p in enumerate(bonus.items):
            assert r_zero[i][0] == r_plain[i][0] and r_zero[i][1].tolist() == r_plain[i][1].tolist(), i
            try:
                ws, wskl = hl.align_s(sc, p, simd=0 if engines == 1 else 1)
                wh = hl.homscore_s(sc, p, simd=0 if engines == 1 else 1)
            except hl.ReferenceUndefined:
                continue
            assert r_bonus[i][0] == ws and r_bonus[i][1].ravel().tolist() == (wskl or []), (engines, vmf, i)
            assert int(h_bonus[i]) == wh, (engines, vmf, i)
            changed += r_bonus[i][0] != r_plain[i][0]
    eng.close()
    assert changed >= 6                                      # the bonus is priced in


@pytest.mark.parametrize("alg,engines", [(0, 1), (1, 2), (2, 0)])
def test_cip_pinned_to_reference_runs(alg, engines):
    """the cp_* fixtures: the reference itself run on a query that carries conserved intron positions (ref_dump -I / -J);
    HomScoreS_ng and alignS_ng under -A0 / -A1 (which read the bonus) and -A2 (which does not) through the ABI"""
    from spaln_amd import engine
    from tests.test_oracle_cip import cip_problem
    eng = engine.Engine(0)
    n = 0
    for path in golden_files("cp_"):
        fx = spdg.load(path)
        sc = spdg.scoring(fx, scalar_engines=engines)
        ps, p = cip_problem(fx)
        assert int(eng.homscore_s(sc, ps)[0]) == int(fx[f"hom_scr_A{alg}"][0]), path
        (scr, skl), = eng.align_s(sc, ps)
        assert scr == int(fx[f"aln_scr_A{alg}"][0]), path
        assert skl.ravel().tolist() == fx[f"aln_skl_A{alg}"].tolist(), path
        # skl_rngS_ng with use_spb(): the intron score of a row with an annotated position carries its bonus (:615)
        fs = fx[f"rng_fstat_A{alg}"]
        (score, fst, ex), = eng.skl_rng_s(sc, ps, [skl], codonk1=fx["prm"]["codonk1"], minl=fx["prm"]["minl"],
                                          jneibr=int(fs[6]), lsg=int(fs[7]))
        assert score == int(fx[f"rng_scr_A{alg}"][0]), path
        assert fst == [int(x) for x in fs[:5]] and ex.tolist() == fx[f"rng_eij_A{alg}"].reshape(-1, 21).tolist(), path
        n += 1
    eng.close()
    assert n == 4
