"""GPU parity of the protein -A1 engines (spdp_h_exact.hip: forwardH1 + Vmf traceback, hirschbergH1) and of
alignH_ng with SpdpScoringH.scalar_engines = 2, against the reference's -A1 goldens and the oracle."""
import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi, synth

pytestmark = pytest.mark.gpu

H_FILES = golden_files("h1_") + golden_files("c1_")      # c1_: dictdisc proteins, species tables (BASELINE config 1)


def _name(f):
    return f.split("/")[-1][:-5]


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


def test_align_a1_goldens(eng):
    """alignH_ng with scalar_engines = 2 against the reference's -A1 run on every protein fixture (a third of
    them differ from both the -A0 and the -A2 result)"""
    for local, fam in ((lo, fa) for lo in (False, True) for fa in (golden_files("h1_"), golden_files("c1_"))):
        cases = [(_name(f), spdg.load(f)) for f in fam if bool(spdg.load(f)["prm"]["local"]) == local]
        if not cases:
            continue
        ref = max((fx for _, fx in cases), key=lambda fx: fx["intpen"].size)
        key = lambda fx: (fx["prm"]["max_vmf_space"], fx["prm"]["ubh"], fx["prm"]["sh"])
        for vmf, ubh, sh in sorted({key(fx) for _, fx in cases}):
            sub = [(n, fx) for n, fx in cases if key(fx) == (vmf, ubh, sh)]
            sc = spdg.scoring_h(ref, scalar_engines=2, max_vmf_space=vmf, ubh=ubh, sh=sh)
            ps = abi.ProblemSetH()
            for _, fx in sub:
                spdg.problem_h(fx, ps)
            res = eng.align_h(sc, ps)
            bad = []
            for (name, fx), (score, skl, flag) in zip(sub, res):
                if flag != 0 or score != int(fx["aln_scr_A1"][0]) or skl.ravel().tolist() != fx["aln_skl_A1"].tolist():
                    bad.append((name, flag, score, int(fx["aln_scr_A1"][0]), skl.ravel().tolist()[:14],
                                fx["aln_skl_A1"].tolist()[:14]))
            assert not bad, bad[:3]


def _subranges(fx, rng, n, m_lo, m_hi):
    q = fx["prm"]
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    ps = abi.ProblemSetH()
    for i in range(n):
        m = int(rng.integers(m_lo, m_hi + 1))
        al = int(rng.integers(0, q["a_right"] - m))
        bl = int(rng.integers(1, 800))
        br = int(rng.integers(max(bl + 3 * m + 300, q["b_right"] - 2500), q["b_right"] + 1))
        exg = (1, 1, 1, 1) if i % 2 else tuple(int(x) for x in rng.integers(0, 2, size=4))
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
               fx["phs5"], fx["phs3"], al, al + m, bl, br, exg, exin=(q["b_left"], q["b_right"]), dinc=dinc)
    return ps


def test_forward_h1_against_oracle(eng):
    """records of spdp_scalar_forward_h under scalar_engines = 2 (forwardH1) on random sub-ranges, ragged stripes"""
    from oracle import oracle
    fx = spdg.load([f for f in H_FILES if f.endswith("h1_400aa.spdg")][0])
    rng = np.random.default_rng(synth.SEED + 95)
    sc = spdg.scoring_h(fx, scalar_engines=2)
    ps = _subranges(fx, rng, 48, 8, 150)
    res = eng.scalar_forward_h(sc, ps)
    bad = []
    for i, (p, (score, skl)) in enumerate(zip(ps.items, res)):
        ws, wskl, wflag = oracle.exact_forward_h(sc, p)
        if wflag:
            continue
        if score != ws or skl.ravel().tolist() != wskl.ravel().tolist():
            bad.append((i, (p.a_left, p.a_right, p.b_left, p.b_right), score, ws, skl.ravel().tolist()[:12],
                        wskl.ravel().tolist()[:12]))
    assert not bad, bad[:3]


def test_hirschberg_h1_against_oracle(eng):
    """cpos rows, ranges and score of spdp_scalar_udh_h under scalar_engines = 2 (hirschbergH1)"""
    from oracle import oracle
    fx = spdg.load([f for f in H_FILES if f.endswith("h1_400aa.spdg")][0])
    rng = np.random.default_rng(synth.SEED + 96)
    sc = spdg.scoring_h(fx, scalar_engines=2)
    for n_im in (1, 3):
        m = 120 + 10 * n_im
        ps = _subranges(fx, rng, 16, m, m)
        scores, cpos, ranges, flags = eng.scalar_udh_h(sc, ps, n_im, (m + n_im) // (n_im + 1))
        bad = []
        for i, p in enumerate(ps.items):
            ws, wcpos, wrng = oracle.exact_udh_h(sc, p, n_im)
            if int(scores[i]) != ws or ranges[i].tolist() != wrng.tolist() or cpos[i].tolist() != wcpos.tolist():
                bad.append((n_im, i, int(scores[i]), ws, ranges[i].tolist(), wrng.tolist(),
                            cpos[i][:2].tolist(), wcpos[:2].tolist()))
        assert not bad, bad[:3]


def test_a1_ladder_against_oracle(eng):
    """alignH_ng under -A1 pushed into the linear-space branches (small MaxVmfSpace) on sub-ranges"""
    from oracle import host_logic_h as hh
    fx = spdg.load([f for f in H_FILES if f.endswith("h1_auto_udh.spdg")][0])
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 97)
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    for vmf, ubh in ((200000, 0), (60000, 3), (20000, 0)):
        sc = spdg.scoring_h(fx, scalar_engines=2, max_vmf_space=vmf, ubh=ubh)
        ps = abi.ProblemSetH()
        for i in range(10):
            al = int(rng.integers(0, 120))
            ar = int(rng.integers(al + 100, min(al + 200, q["a_right"]) + 1))
            bl = int(rng.integers(1, 500))
            br = int(rng.integers(q["b_right"] - 1500, q["b_right"] + 1))
            exg = (1, 1, 1, 1) if i % 2 else tuple(int(x) for x in rng.integers(0, 2, size=4))
            ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
                   fx["phs5"], fx["phs3"], al, ar, bl, br, exg, exin=(q["b_left"], q["b_right"]), dinc=dinc)
        res = eng.align_h(sc, ps)
        bad, n_ok = [], 0
        for i, (p, (score, skl, flag)) in enumerate(zip(ps.items, res)):
            try:
                ws, wskl = hh.align_h(sc, p, simd=1)
            except (hh.ReferenceUndefined, hh.NotRestated):
                assert flag != 0, i
                continue
            n_ok += 1
            if flag != 0 or score != ws or skl.ravel().tolist() != (wskl or []):
                bad.append((vmf, i, (p.a_left, p.a_right, p.b_left, p.b_right), flag, score, ws,
                            skl.ravel().tolist()[:12], (wskl or [])[:12]))
        assert n_ok >= 7 and not bad, bad[:3]


def test_pipelined_stripes_equal_one_group(eng, monkeypatch):
    """spdh_exact with the stripes of a problem as a pipeline of waves (the default; 1, 2 or 4 problems per wave) against the
    same kernel with one 16-lane group per problem (SPDP_HX_PIPE=0, the form that carries the reference's stale link planes
    from stripe to stripe): records, scores, cpos rows and ranges of random sub-ranges, the ones without a path included
    (those meet a poisoned link in the pipelined form and are run again)"""
    fx = spdg.load([f for f in H_FILES if f.endswith("h1_400aa.spdg")][0])
    sc = spdg.scoring_h(fx, scalar_engines=2)
    rng = np.random.default_rng(synth.SEED + 98)
    ps = _subranges(fx, rng, 96, 20, 330)

    def run():
        fwd = [(s, k.ravel().tolist()) for s, k in eng.scalar_forward_h(sc, ps)]
        udh = {}
        for n_im in (1, 2):
            sub = abi.ProblemSetH()
            sub.items = [p for p in ps.items if p.a_right - p.a_left >= 64]
            s, c, r, f = eng.scalar_udh_h(sc, sub, n_im, 40)
            udh[n_im] = (s.tolist(), c.tolist(), r.tolist())
        return fwd, udh

    monkeypatch.setenv("SPDP_HX_PIPE", "0")
    want = run()
    monkeypatch.delenv("SPDP_HX_PIPE")
    for groups in ("1", "2", "4"):
        monkeypatch.setenv("SPDP_HX_GROUPS", groups)
        got = run()
        assert got[0] == want[0], groups
        assert got[1] == want[1], groups
