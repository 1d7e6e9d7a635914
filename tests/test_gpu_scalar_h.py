"""GPU parity of the scalar aa x genome engine (spdp_scalar_forward_h = forwardH_ng + trcbkalignH_ng's
record hand-over, spdp_h_rowwave.hip) and of the below-8-rows branch of the dispatch, against the
reference's -A0 goldens and the oracle."""
import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi, synth

pytestmark = pytest.mark.gpu

H_FILES = golden_files("h1_") + golden_files("c1_")      # c1_: dictdisc proteins, species tables (BASELINE config 1)
FAMILIES = [golden_files("h1_"), golden_files("c1_")]    # one parameter set (scoring) per family


def _name(f):
    return f.split("/")[-1][:-5]


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


def _batches(local):
    for fam in FAMILIES:
        cases = [(_name(f), spdg.load(f)) for f in fam if bool(spdg.load(f)["prm"]["local"]) == local]
        if not cases:
            continue
        sc = spdg.scoring_h(max((fx for _, fx in cases), key=lambda fx: fx["intpen"].size))
        ps = abi.ProblemSetH()
        for _, fx in cases:
            spdg.problem_h(fx, ps)
        yield cases, sc, ps


@pytest.mark.parametrize("local", [False, True])
def test_scores_against_reference_a0(eng, local):
    """HomScoreH_ng under -A0 = forwardH_ng without a Vmf: every fixture, one batch"""
    for cases, sc, ps in _batches(local):
        res = eng.scalar_forward_h(sc, ps, traceback=False)
        bad = [(name, s, int(fx["hom_scr_A0"][0])) for (name, fx), (s, _) in zip(cases, res)
               if s != int(fx["hom_scr_A0"][0])]
        assert not bad, bad


@pytest.mark.parametrize("local", [False, True])
def test_records_against_oracle(eng, local):
    """raw Mfile records of trcbkalignH_ng's scalar branch (the oracle is pinned on the -A0 alignments)"""
    from oracle import oracle
    for cases, sc, ps in _batches(local):
        res = eng.scalar_forward_h(sc, ps)
        bad = []
        for (name, fx), p, (s, skl) in zip(cases, ps.items, res):
            ws, wskl = oracle.scalar_forward_h(sc, p)
            if s != ws or skl.tolist() != wskl.tolist():
                bad.append((name, s, ws, skl.ravel().tolist()[:12], wskl.ravel().tolist()[:12]))
        assert not bad, bad[:4]


@pytest.mark.parametrize("alg", [2, 3])
def test_below_8_rows_goldens(eng, alg):
    """-A2 / -A3 on queries below 8 residues: alignH_ng and HomScoreH_ng go through the scalar engine"""
    cases = [(m, spdg.load([f for f in H_FILES if _name(f) == f"h1_tiny_m{m}"][0])) for m in (3, 5, 7)]
    sc = spdg.scoring_h(cases[0][1], nquant=None if alg == 2 else 1)
    ps = abi.ProblemSetH()
    for _, fx in cases:
        spdg.problem_h(fx, ps)
    res = eng.align_h(sc, ps)
    hom = eng.homscore_h(sc, ps)
    for (m, fx), (score, skl, flag), hs in zip(cases, res, hom):
        assert flag == 0 and score == int(fx[f"aln_scr_A{alg}"][0]), m
        assert skl.ravel().tolist() == fx[f"aln_skl_A{alg}"].tolist(), m
        assert int(hs) == int(fx[f"hom_scr_A{alg}"][0]), m


def test_small_subranges_against_oracle(eng):
    """random sub-ranges with 1 .. 7 query rows and all end-gap flag combinations: engine and dispatch"""
    from oracle import oracle, host_logic_h as hh
    fx = spdg.load([f for f in H_FILES if f.endswith("h1_400aa.spdg")][0])
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 81)
    sc = spdg.scoring_h(fx)
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    ps = abi.ProblemSetH()
    for i in range(160):
        m = int(rng.integers(1, 8))
        al = int(rng.integers(0, q["a_right"] - m))
        bl = int(rng.integers(1, q["b_right"] - 700))
        br = bl + int(rng.integers(max(3 * m + 2, 12), 600))
        exg = tuple(int(x) for x in rng.integers(0, 3 if i % 4 == 0 else 2, size=4))
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
               fx["phs5"], fx["phs3"], al, al + m, bl, br, exg, exin=(q["b_left"], q["b_right"]), dinc=dinc)
    res = eng.scalar_forward_h(sc, ps)
    bad = []
    for i, (p, (s, skl)) in enumerate(zip(ps.items, res)):
        ws, wskl = oracle.scalar_forward_h(sc, p)
        if s != ws or skl.tolist() != wskl.tolist():
            bad.append((i, (p.a_left, p.a_right, p.b_left, p.b_right), (p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr),
                        s, ws, skl.ravel().tolist()[:10], wskl.ravel().tolist()[:10]))
    assert not bad, bad[:4]
    # the same problems through alignH_ng
    res = eng.align_h(sc, ps)
    n_ok = 0
    for i, (p, (score, skl, flag)) in enumerate(zip(ps.items, res)):
        try:
            ws, wskl = hh.align_h(sc, p)
        except hh.NotRestated:
            assert flag == 1, i
            continue
        n_ok += 1
        if flag != 0 or score != ws or skl.ravel().tolist() != (wskl or []):
            bad.append((i, (p.a_left, p.a_right, p.b_left, p.b_right), flag, score, ws,
                        skl.ravel().tolist()[:12], (wskl or [])[:12]))
    assert n_ok >= 100 and not bad, bad[:4]


# ---- -A0 mode: hirschbergH_ng and the scalar ladder ---------------------------------------------
def test_align_a0_goldens(eng):
    """alignH_ng / HomScoreH_ng with SpdpScoringH.scalar_engines = 1 against the reference's -A0 output"""
    for local, fam in ((lo, fa) for lo in (False, True) for fa in FAMILIES):
        cases = [(_name(f), spdg.load(f)) for f in fam if bool(spdg.load(f)["prm"]["local"]) == local]
        if not cases:
            continue
        ref = max((fx for _, fx in cases), key=lambda fx: fx["intpen"].size)
        key = lambda fx: (fx["prm"]["max_vmf_space"], fx["prm"]["ubh"], fx["prm"]["sh"])
        for vmf, ubh, sh in sorted({key(fx) for _, fx in cases}):
            sub = [(n, fx) for n, fx in cases if key(fx) == (vmf, ubh, sh)]
            sc = spdg.scoring_h(ref, scalar_engines=1, max_vmf_space=vmf, ubh=ubh, sh=sh)
            ps = abi.ProblemSetH()
            for _, fx in sub:
                spdg.problem_h(fx, ps)
            res = eng.align_h(sc, ps)
            hom = eng.homscore_h(sc, ps)
            bad = []
            for (name, fx), (score, skl, flag), hs in zip(sub, res, hom):
                if (flag != 0 or score != int(fx["aln_scr_A0"][0]) or skl.ravel().tolist() != fx["aln_skl_A0"].tolist()
                        or int(hs) != int(fx["hom_scr_A0"][0])):
                    bad.append((name, flag, score, int(fx["aln_scr_A0"][0]), skl.ravel().tolist()[:14],
                                fx["aln_skl_A0"].tolist()[:14]))
            assert not bad, bad[:3]


def test_hirschberg_h_ng_against_oracle(eng):
    """cpos rows (incl. the diagonal bounds), ranges and score of spdp_scalar_udh_h on sub-ranges"""
    from oracle import oracle
    fx = spdg.load([f for f in H_FILES if f.endswith("h1_400aa.spdg")][0])
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 93)
    sc = spdg.scoring_h(fx, scalar_engines=1)
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    for n_im in (1, 3):
        ps = abi.ProblemSetH()
        m = 120 + 10 * n_im
        intvl = (m + n_im) // (n_im + 1)
        for i in range(16):
            al = int(rng.integers(0, q["a_right"] - m))
            bl = int(rng.integers(1, 800))
            br = int(rng.integers(max(bl + 3 * m + 300, q["b_right"] - 2500), q["b_right"] + 1))
            exg = (1, 1, 1, 1) if i % 2 else tuple(int(x) for x in rng.integers(0, 2, size=4))
            ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
                   fx["phs5"], fx["phs3"], al, al + m, bl, br, exg, exin=(q["b_left"], q["b_right"]), dinc=dinc)
        scores, cpos, ranges, flags = eng.scalar_udh_h(sc, ps, n_im, intvl)
        bad = []
        for i, p in enumerate(ps.items):
            ws, wcpos, wrng, wflag = oracle.scalar_udh_h(sc, p, n_im, intvl)
            ok = int(flags[i]) == wflag
            if wflag == 0:
                ok = ok and int(scores[i]) == ws and ranges[i].tolist() == wrng.tolist() and cpos[i].tolist() == wcpos.tolist()
            if not ok:
                bad.append((n_im, i, int(scores[i]), ws, int(flags[i]), wflag, ranges[i].tolist(), wrng.tolist(),
                            cpos[i][:2].tolist(), wcpos[:2].tolist()))
        assert not bad, bad[:3]


def test_a0_ladder_against_oracle(eng):
    """alignH_ng under -A0 pushed into the linear-space branches (small MaxVmfSpace) on sub-ranges"""
    from oracle import host_logic_h as hh
    fx = spdg.load([f for f in H_FILES if f.endswith("h1_auto_udh.spdg")][0])
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 94)
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    for vmf, ubh in ((200000, 0), (60000, 3), (20000, 0)):
        sc = spdg.scoring_h(fx, scalar_engines=1, max_vmf_space=vmf, ubh=ubh)
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
                ws, wskl = hh.align_h(sc, p, simd=0)
            except (hh.ReferenceUndefined, hh.NotRestated):
                assert flag == 1, i
                continue
            n_ok += 1
            if flag != 0 or score != ws or skl.ravel().tolist() != (wskl or []):
                bad.append((vmf, i, (p.a_left, p.a_right, p.b_left, p.b_right), flag, score, ws,
                            skl.ravel().tolist()[:12], (wskl or [])[:12]))
        assert n_ok >= 7 and not bad, bad[:3]


def test_record_budget_retry_and_groups(eng, monkeypatch):
    """forwardH_ng / forwardH1 with Vmf records: a problem that outgrows its record budget is run again with a larger one
    (test hook SPDP_VMF_TEST_TINY: every first budget is 96 records), and a batch is cut into launches by record memory"""
    fx = spdg.load([f for f in H_FILES if f.endswith("h1_400aa.spdg")][0])
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 97)
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    for sel in (1, 2):
        sc = spdg.scoring_h(fx, scalar_engines=sel)
        ps = abi.ProblemSetH()
        for i in range(24):
            m = int(rng.integers(40, 200))
            al = int(rng.integers(0, q["a_right"] - m))
            bl = int(rng.integers(1, 600))
            br = int(rng.integers(max(bl + 3 * m + 200, q["b_right"] - 1500), q["b_right"] + 1))
            ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
                   fx["phs5"], fx["phs3"], al, al + m, bl, br, (1, 1, 1, 1), exin=(q["b_left"], q["b_right"]), dinc=dinc)
        want = [(s, skl.tolist(), f) for s, skl, f in eng.align_h(sc, ps)]
        monkeypatch.setenv("SPDP_VMF_TEST_TINY", "1")
        got = [(s, skl.tolist(), f) for s, skl, f in eng.align_h(sc, ps)]
        monkeypatch.delenv("SPDP_VMF_TEST_TINY")
        assert want == got, sel
        monkeypatch.setenv("SPDP_VMF_GB", "1")          # (1 GiB: several launches for the -A0 ladder's slabs is not guaranteed; the path runs)
        again = [(s, skl.tolist(), f) for s, skl, f in eng.align_h(sc, ps)]
        monkeypatch.delenv("SPDP_VMF_GB")
        assert want == again, sel


@pytest.mark.parametrize("engine_kind", ["a0", "a1"])
def test_ambiguous_bases_against_oracle(eng, engine_kind):
    """tron codes turned into AMB at random (1 - 3 % of the window; signals kept): the codon an intron splits is then
    often half-defined, and the engines must price it as SpJunc::spjseq does -- a codon stands when its own three bases
    do (the oracle's rule is pinned to the reference's function, tests/test_oracle_spjseq.py)"""
    from oracle import oracle
    rng = np.random.default_rng(4242)
    names = ("h1_basic", "h1_400aa", "h1_divergent", "h1_frameshift", "h1_query_indel", "h1_amb_junction0")
    cases = [spdg.load([f for f in H_FILES if _name(f) == n][0]) for n in names]
    sc = spdg.scoring_h(max(cases, key=lambda fx: fx["intpen"].size))
    ps = abi.ProblemSetH()
    for fx in cases:
        for frac in (0.01, 0.03):
            fx2 = dict(fx)
            b = np.array(fx["b_codes"], dtype=np.uint8)
            hit = rng.random(b.size) < frac
            b[hit] = 2                                            # AMB
            fx2["b_codes"] = b
            spdg.problem_h(fx2, ps)
    sc.scalar_engines = 1 if engine_kind == "a0" else 2
    got = eng.scalar_forward_h(sc, ps)
    n_cmp = 0
    for p, (s, skl) in zip(ps.items, got):
        if engine_kind == "a0":
            ws, wskl = oracle.scalar_forward_h(sc, p)
        else:
            ws, wskl, flag = oracle.exact_forward_h(sc, p)
            if flag:
                continue
        assert s == ws and skl.ravel().tolist() == np.asarray(wskl).ravel().tolist()
        n_cmp += 1
    assert n_cmp >= 8
