"""GPU parity of the -A0 mode of the cDNA path: the scalar linear-space engine (spdp_scalar_udh =
hirschbergS_ng) against the oracle, and alignS_ng with SpdpScoring.scalar_engines = 1 end to end against
the reference's -A0 alignments (tests/golden/s1_*.spdg)."""
import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi, synth

pytestmark = pytest.mark.gpu

S_FILES = golden_files("s1_")


def _name(f):
    return f.split("/")[-1][:-5]


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


def test_align_a0_goldens(eng):
    """every cDNA fixture, one batch per mode: score + final corner list as the reference prints under -A0"""
    for local in (False, True):
        cases = [(_name(f), spdg.load(f)) for f in S_FILES if ("local" in _name(f)) == local]
        sc = spdg.scoring(max((fx for _, fx in cases), key=lambda fx: fx["intpen"].size), scalar_engines=1)
        key = lambda fx: (fx["prm"]["max_vmf_space"], fx["prm"]["ubh"], fx["prm"]["sh"])
        for vmf, ubh, sh in sorted({key(fx) for _, fx in cases}):
            if True:
                sub = [(n, fx) for n, fx in cases if key(fx) == (vmf, ubh, sh)]
                sc.max_vmf_space, sc.ubh, sc.sh = vmf, ubh, sh
                ps = abi.ProblemSet()
                for _, fx in sub:
                    spdg.problem(fx, ps)
                res = eng.align_s(sc, ps)
                hom = eng.homscore_s(sc, ps)
                bad = []
                for (name, fx), (score, skl), hs in zip(sub, res, hom):
                    if (score != int(fx["aln_scr_A0"][0]) or skl.ravel().tolist() != fx["aln_skl_A0"].tolist()
                            or int(hs) != int(fx["hom_scr_A0"][0])):
                        bad.append((name, score, int(fx["aln_scr_A0"][0]), skl.ravel().tolist()[:14],
                                    fx["aln_skl_A0"].tolist()[:14]))
                assert not bad, bad[:3]


def test_hirschberg_s_ng_against_oracle(eng):
    """cpos rows (incl. the diagonal bounds in [8], [9]), ranges and score on sub-ranges of a fixture"""
    from oracle import oracle
    fx = spdg.load([f for f in S_FILES if f.endswith("s1_1400nt.spdg")][0])
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 91)
    sc = spdg.scoring(fx, scalar_engines=1)
    extra = dict(cano5=fx["cano5"], cano3=fx["cano3"],
                 dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
    for n_im in (1, 4):
        ps = abi.ProblemSet()
        intvl = None
        for i in range(16):
            m = 420 + 14 * n_im                       # same imd_intvl for the whole batch
            al = int(rng.integers(0, q["a_right"] - m))
            bl = int(rng.integers(0, 600))
            br = int(rng.integers(max(bl + m + 200, q["b_right"] - 900), q["b_right"] + 1))
            exg = (1, 1, 1, 1) if i % 2 else tuple(int(x) for x in rng.integers(0, 2, size=4))
            ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, al + m, bl, br, exg, **extra)
            intvl = (m + n_im) // (n_im + 1)
        scores, cpos, ranges, flags = eng.scalar_udh(sc, ps, n_im, intvl)
        bad = []
        for i, p in enumerate(ps.items):
            ws, wcpos, wrng, wflag = oracle.scalar_udh(sc, p, n_im, intvl)
            ok = int(flags[i]) == wflag
            if wflag == 0:
                ok = ok and int(scores[i]) == ws and ranges[i].tolist() == wrng.tolist() and cpos[i].tolist() == wcpos.tolist()
            if not ok:
                bad.append((n_im, i, int(scores[i]), ws, int(flags[i]), wflag, ranges[i].tolist(), wrng.tolist(),
                            cpos[i][:2].tolist(), wcpos[:2].tolist()))
        assert not bad, bad[:3]


def test_a0_ladder_against_oracle(eng):
    """alignS_ng under -A0 pushed into the recurrent and recursive linear-space branches on sub-ranges"""
    from oracle import host_logic as hl
    fx = spdg.load([f for f in S_FILES if f.endswith("s1_auto_udh.spdg")][0])
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 92)
    extra = dict(cano5=fx["cano5"], cano3=fx["cano3"],
                 dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
    for vmf, ubh in ((600000, 0), (100000, 3), (40000, 0)):
        sc = spdg.scoring(fx, scalar_engines=1, max_vmf_space=vmf, ubh=ubh)
        ps = abi.ProblemSet()
        for i in range(12):
            al = int(rng.integers(0, 400))
            ar = int(rng.integers(al + 300, min(al + 600, q["a_right"]) + 1))
            bl = int(rng.integers(0, 500))
            br = int(rng.integers(q["b_right"] - 1500, q["b_right"] + 1))
            exg = (1, 1, 1, 1) if i % 2 else tuple(int(x) for x in rng.integers(0, 2, size=4))
            ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, ar, bl, br, exg, **extra)
        res = eng.align_s(sc, ps, allow_partial=True)
        bad, n_ok = [], 0
        for i, (p, (score, skl)) in enumerate(zip(ps.items, res)):
            try:
                ws, wskl = hl.align_s(sc, p, simd=0)
            except hl.ReferenceUndefined:
                assert len(skl) == 0, i
                continue
            n_ok += 1
            if score != ws or skl.ravel().tolist() != (wskl or []):
                bad.append((vmf, i, (p.a_left, p.a_right, p.b_left, p.b_right), score, ws,
                            skl.ravel().tolist()[:12], (wskl or [])[:12]))
        assert n_ok >= 9 and not bad, bad[:3]
