"""GPU parity of the aa x genome path (libspdp_hip.so through its C ABI) against the reference's
goldens (tests/golden/h1_*.spdg) and, on seeded synthetic loci, against the oracle."""
import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi, synth

pytestmark = pytest.mark.gpu

H_FILES = golden_files("h1_")
UNDEFINED = {"h1_cut_right", "h1_random"}     # the reference starts its traceback outside its bitmap
LOCAL = {"h1_local", "h1_local_udh"}          # -LS: its own kernel variants, run separately
BELOW_8 = {"h1_tiny_m3", "h1_tiny_m5", "h1_tiny_m7"}   # every dispatch sends these to the scalar engine


def _name(f):
    return f.split("/")[-1][:-5]


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


def _cases(tag, wip_only=False):
    out = []
    for f in H_FILES:
        if _name(f) in LOCAL or (wip_only and _name(f) in BELOW_8):
            continue
        fx = spdg.load(f)
        out.append((_name(f), fx))
    return out


@pytest.mark.parametrize("tag", ["qn", "q1"])
def test_forward_h1_wip_goldens(eng, tag):
    """all fixtures in one batch per intron model: raw score + raw corner records"""
    cases = _cases(tag, wip_only=True)
    sc = spdg.scoring_h(cases[0][1], nquant=None if tag == "qn" else 1)
    ps = abi.ProblemSetH()
    for _, fx in cases:
        spdg.problem_h(fx, ps)
    res = eng.wip_forward_h(sc, ps)
    bad = []
    for (name, fx), (score, skl, flag) in zip(cases, res):
        ok = score == int(fx[f"wip_{tag}_fwd_scr"][0])
        if name in UNDEFINED:
            ok = ok and flag == -2
        else:
            ok = ok and flag == 0 and skl.ravel().tolist() == fx[f"wip_{tag}_fwd_skl"].tolist()
        if not ok:
            bad.append((name, score, flag, skl.ravel().tolist()[:12], fx[f"wip_{tag}_fwd_skl"].tolist()[:12]))
    assert not bad, bad


@pytest.mark.parametrize("alg", [2, 3])
def test_align_h_goldens(eng, alg):
    all_cases = _cases(alg)
    key = lambda fx: (fx["prm"]["max_vmf_space"], fx["prm"]["ubh"], fx["prm"]["sh"])
    bad = []
    for vmf, ubh, sh in sorted({key(fx) for _, fx in all_cases}):       # one batch per ladder setting
        cases = [(n, fx) for n, fx in all_cases if key(fx) == (vmf, ubh, sh)]
        sc = spdg.scoring_h(all_cases[0][1], nquant=None if alg == 2 else 1, max_vmf_space=vmf, ubh=ubh, sh=sh)
        ps = abi.ProblemSetH()
        for _, fx in cases:
            spdg.problem_h(fx, ps)
        res = eng.align_h(sc, ps)
        hom = eng.homscore_h(sc, ps)
        for (name, fx), (score, skl, flag), hs in zip(cases, res, hom):
            ok = score == int(fx[f"aln_scr_A{alg}"][0]) and int(hs) == int(fx[f"hom_scr_A{alg}"][0])
            if name in UNDEFINED:
                ok = ok and flag == -2
            else:
                ok = ok and flag == 0 and skl.ravel().tolist() == fx[f"aln_skl_A{alg}"].tolist()
            if not ok:
                bad.append((name, score, flag, skl.ravel().tolist()[:14], fx[f"aln_skl_A{alg}"].tolist()[:14]))
    assert not bad, bad


@pytest.mark.parametrize("tag", ["qn", "q1"])
@pytest.mark.parametrize("name", sorted(LOCAL))
def test_local_mode_golden(eng, tag, name):
    """-LS fixtures: the forward engine, and alignH_ng (h1_local_udh: through hirschbergH1_wip with local ends,
    which ends the alignment one row beyond the query as the reference does)"""
    fx = spdg.load([f for f in H_FILES if _name(f) == name][0])
    sc = spdg.scoring_h(fx, nquant=None if tag == "qn" else 1)
    ps, _ = spdg.problem_h(fx)
    (score, skl, flag), = eng.wip_forward_h(sc, ps)
    assert flag == 0 and score == int(fx[f"wip_{tag}_fwd_scr"][0])
    assert skl.ravel().tolist() == fx[f"wip_{tag}_fwd_skl"].tolist()
    alg = 2 if tag == "qn" else 3
    (score, skl, flag), = eng.align_h(sc, ps)
    assert flag == 0 and score == int(fx[f"aln_scr_A{alg}"][0])
    assert skl.ravel().tolist() == fx[f"aln_skl_A{alg}"].tolist()
    assert int(eng.homscore_h(sc, ps)[0]) == int(fx[f"hom_scr_A{alg}"][0])


def test_local_mode_against_oracle(eng):
    """local ends on sub-ranges (LocalL / LocalR only hold when both sequences are free on that side)"""
    from oracle import oracle
    fx = spdg.load([f for f in H_FILES if f.endswith("h1_400aa.spdg")][0])
    sc = spdg.scoring_h(fx, local=1)
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 78)
    ps = abi.ProblemSetH()
    for i in range(48):
        al = int(rng.integers(0, 150))
        ar = int(rng.integers(al + 30, q["a_right"] + 1))
        bl = int(rng.integers(0, 1200))
        br = int(rng.integers(max(bl + 3 * (ar - al) // 2, bl + 300), q["b_right"] + 1))
        exg = (1, 1, 1, 1) if i % 3 else tuple(int(x) for x in rng.integers(0, 2, size=4))
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
               fx["phs5"], fx["phs3"], al, ar, bl, br, exg, exin=(q["b_left"], q["b_right"]))
    res = eng.wip_forward_h(sc, ps)
    bad = []
    for i, (p, (score, skl, flag)) in enumerate(zip(ps.items, res)):
        s, oskl, oflag = oracle.wip_forward_h(sc, p)
        ok = score == s and flag == {0: 0, -2: -1, -3: -2}[oflag]
        if oflag == 0:
            ok = ok and skl.tolist() == oskl.tolist()
        if not ok:
            bad.append((i, (p.a_left, p.a_right, p.b_left, p.b_right), score, s, flag, oflag,
                        skl.ravel().tolist()[:10], oskl.ravel().tolist()[:10]))
    assert not bad, bad[:4]


def test_loaded_gpu_against_oracle(eng):
    """many waves per CU at once: goldens' signals re-used on windows cut at many offsets"""
    from oracle import oracle
    fx = spdg.load([f for f in H_FILES if f.endswith("h1_400aa.spdg")][0])
    sc = spdg.scoring_h(fx)
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 77)
    ps = abi.ProblemSetH()
    for i in range(96):
        al = int(rng.integers(0, 120))
        ar = int(rng.integers(al + 40, q["a_right"] + 1))
        bl = int(rng.integers(0, 900))
        br = int(rng.integers(max(bl + 3 * (ar - al) // 2, bl + 300), q["b_right"] + 1))
        exg = tuple(int(x) for x in rng.integers(0, 2, size=4))
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
               fx["phs5"], fx["phs3"], al, ar, bl, br, exg, exin=(q["b_left"], q["b_right"]))
    res = eng.wip_forward_h(sc, ps)
    bad = []
    for i, (p, (score, skl, flag)) in enumerate(zip(ps.items, res)):
        s, oskl, oflag = oracle.wip_forward_h(sc, p)
        want_flag = {0: 0, -2: -1, -3: -2}[oflag]
        ok = score == s and flag == want_flag
        if oflag == 0:
            ok = ok and skl.tolist() == oskl.tolist()
        if not ok:
            bad.append((i, (p.a_left, p.a_right, p.b_left, p.b_right), score, s, flag, oflag,
                        skl.ravel().tolist()[:10], oskl.ravel().tolist()[:10]))
    assert not bad, bad[:4]


def _udh_cases():
    import re
    out = []
    for f in H_FILES:
        fx = spdg.load(f)                       # h1_local too: hirschbergH1_wip with local ends (spdh_local_udh)
        for k in fx:
            m = re.match(r"wip_(qn|q1)_udh(\d+)_scr", k)
            if m:
                out.append((f, m.group(1), int(m.group(2))))
    return out


def test_hirschberg_h1_wip_goldens(eng):
    """score, cpos rows and written-back ranges of hirschbergH1_wip, every (case, model, n_im)"""
    bad = []
    for path, tag, n_im in _udh_cases():
        fx = spdg.load(path)
        sc = spdg.scoring_h(fx, nquant=None if tag == "qn" else 1)
        ps, _ = spdg.problem_h(fx)
        scores, cpos, ranges = eng.wip_udh_h(sc, ps, n_im)
        ok = (int(scores[0]) == int(fx[f"wip_{tag}_udh{n_im}_scr"][0])
              and cpos[0].ravel().tolist() == fx[f"wip_{tag}_udh{n_im}_cpos"].tolist()
              and ranges[0].tolist() == fx[f"wip_{tag}_udh{n_im}_rng"][:4].tolist())
        if not ok:
            bad.append((_name(path), tag, n_im, int(scores[0]), ranges[0].tolist(), cpos[0][:2].tolist(),
                        fx[f"wip_{tag}_udh{n_im}_cpos"][:20].tolist()))
    assert not bad, bad[:3]


def test_udh_ladder_against_oracle(eng):
    """alignH_ng forced into the linear-space branch (small MaxVmfSpace) on sub-ranges, vs the oracle ladder"""
    from oracle import oracle, host_logic_h as hh
    fx = spdg.load([f for f in H_FILES if f.endswith("h1_450aa_auto.spdg")][0])
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 79)
    for vmf, ubh in ((1500000, 0), (600000, 0), (300000, 3), (150000, 0)):
        sc = spdg.scoring_h(fx, max_vmf_space=vmf, ubh=ubh)
        ps = abi.ProblemSetH()
        for i in range(24):
            al = int(rng.integers(0, 100))
            ar = int(rng.integers(al + 200, q["a_right"] + 1))
            bl = int(rng.integers(0, 500))
            br = int(rng.integers(max(bl + 3 * (ar - al), q["b_right"] - 1500), q["b_right"] + 1))
            exg = (1, 1, 1, 1) if i % 2 else tuple(int(x) for x in rng.integers(0, 2, size=4))
            ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
                   fx["phs5"], fx["phs3"], al, ar, bl, br, exg, exin=(q["b_left"], q["b_right"]))
        res = eng.align_h(sc, ps)
        bad = []
        n_udh = sum(2.0 * (p.a_right - p.a_left) * (p.b_right - p.b_left + 3 * (p.a_right - p.a_left)) >= vmf
                    for p in ps.items)
        assert n_udh >= 20                                   # the linear-space branch is what runs
        assert sum(f == 0 for _, _, f in res) >= 16
        for i, (p, (score, skl, flag)) in enumerate(zip(ps.items, res)):
            try:
                ws, wskl = hh.align_h(sc, p)
                wflag = 0
            except hh.ReferenceUndefined:
                ws, wskl, wflag = None, None, -2
            except hh.ReferenceFatal:
                ws, wskl, wflag = None, None, -1
            except hh.NotRestated:
                ws, wskl, wflag = None, None, 1
            ok = flag == wflag
            if wflag == 0:
                ok = ok and score == ws and skl.ravel().tolist() == (wskl or [])
            if not ok:
                bad.append((vmf, i, (p.a_left, p.a_right, p.b_left, p.b_right), flag, wflag, score, ws,
                            skl.ravel().tolist()[:12], (wskl or [])[:12]))
        assert not bad, bad[:3]


def test_skl_rng_h_goldens(eng):
    """spdp_skl_rng_h (skl_rngH_ng on the device) on the reference's own corner lists"""
    bad, n_checked = [], 0
    for f in H_FILES:
        if _name(f) == "h1_cut_right":
            continue                                  # alignment beyond the window: reference reads its heap
        fx = spdg.load(f)
        h = dict(zip(spdg.HPARAM_NAMES, (int(x) for x in fx["hparams"])))
        rp = [int(x) for x in fx["rparams"]]
        for alg in (0, 2, 3):
            if f"rng_eij_A{alg}" not in fx:
                continue
            if _name(f) == "h1_local_udh" and alg in (2, 3):
                continue                              # alignment ends one row beyond the query (reference quirk)
            sc = spdg.scoring_h(fx, nquant=None if alg != 3 else 1)
            ps, _ = spdg.problem_h(fx)
            (score, fst, ex), = eng.skl_rng_h(sc, ps, [fx[f"aln_skl_A{alg}"].reshape(-1, 2)], minl=fx["prm"]["minl"],
                                               jneibr=rp[4], lcl=h["lcl"], sup_tcodon=rp[1])
            ok = (score == int(fx[f"rng_scr_A{alg}"][0]) and fst == [int(x) for x in fx[f"rng_fstat_A{alg}"][:5]]
                  and ex.tolist() == fx[f"rng_eij_A{alg}"].reshape(-1, 21).tolist())
            n_checked += 1
            if not ok:
                bad.append((_name(f), alg, score, int(fx[f"rng_scr_A{alg}"][0]), fst, fx[f"rng_fstat_A{alg}"][:5].tolist()))
    assert n_checked >= 60 and not bad, bad[:4]


def test_skl_edits_h_goldens(eng):
    """spdp_skl_edits_h: the Cigar and (post-processed) Vulgar edit records of skl_rngH_ng on the reference's own
    corner lists, record for record -- h1_* and the dictdisc c1_* fixtures, -A2 and -A0 alignments"""
    from spaln_amd import abi as _abi
    from tests.conftest import golden_files as _gf
    bad, n_checked, n_introns = [], 0, 0
    for f in H_FILES + _gf("c1_"):
        if _name(f) == "h1_cut_right":
            continue
        fx = spdg.load(f)
        h = dict(zip(spdg.HPARAM_NAMES, (int(x) for x in fx["hparams"])))
        rp = [int(x) for x in fx["rparams"]]
        for alg in (0, 2):
            if f"rng_cigar_A{alg}" not in fx:
                continue
            if _name(f) == "h1_local_udh" and alg == 2:
                continue
            sc = spdg.scoring_h(fx)
            ps, _ = spdg.problem_h(fx)
            kw = dict(minl=fx["prm"]["minl"], jneibr=rp[4], lcl=h["lcl"], sup_tcodon=rp[1])
            skl = [fx[f"aln_skl_A{alg}"].reshape(-1, 2)]
            cig, = eng.skl_edits_h(sc, ps, skl, _abi.FMT_CIGAR, **kw)
            vul, = eng.skl_edits_h(sc, ps, skl, _abi.FMT_VULGAR, **kw)
            ok = (cig[:, :2].ravel().tolist() == fx[f"rng_cigar_A{alg}"].tolist()
                  and vul.ravel().tolist() == fx[f"rng_vulgar_A{alg}"].tolist())
            n_checked += 1
            n_introns += int((cig[:, 0] == ord("N")).sum())
            if not ok:
                bad.append((_name(f), alg, cig[:8, :2].ravel().tolist(), fx[f"rng_cigar_A{alg}"][:16].tolist(),
                            vul[:6].ravel().tolist(), fx[f"rng_vulgar_A{alg}"][:18].tolist()))
    assert n_checked >= 50 and n_introns >= 60 and not bad, bad[:3]


def test_align_h_a6_recursive_switch(eng):
    """SpdpScoringH.recursive (algmode.alg & 4): lspH_ng's recursive branch, against the reference's -A6 output"""
    cases = [(n, fx) for n, fx in _cases(3) if n not in UNDEFINED]
    key = lambda fx: (fx["prm"]["max_vmf_space"], fx["prm"]["ubh"], fx["prm"]["sh"])
    for vmf, ubh, sh in sorted({key(fx) for _, fx in cases}):
        sub = [(n, fx) for n, fx in cases if key(fx) == (vmf, ubh, sh)]
        sc = spdg.scoring_h(max((fx for _, fx in sub), key=lambda fx: fx["intpen"].size), nquant=1, recursive=1,
                            max_vmf_space=vmf, ubh=ubh, sh=sh)
        ps = abi.ProblemSetH()
        for _, fx in sub:
            spdg.problem_h(fx, ps)
        res = eng.align_h(sc, ps)
        for (name, fx), (score, skl, flag) in zip(sub, res):
            assert flag == 0 and score == int(fx["aln_scr_A6"][0]), name
            assert skl.ravel().tolist() == fx["aln_skl_A6"].tolist(), name


def test_local_udh_against_oracle(eng):
    """hirschbergH1_wip with local ends (-LS, spdh_local_udh) on random sub-ranges against the oracle (pinned by
    the h1_local records), and alignH_ng in local mode pushed into the linear-space branches"""
    from oracle import oracle, host_logic_h as hh
    fx = spdg.load([f for f in H_FILES if f.endswith("h1_local.spdg")][0])
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 99)
    sc = spdg.scoring_h(fx)
    assert sc.local
    for n_im in (1, 3):
        ps = abi.ProblemSetH()
        for i in range(24):
            m = int(rng.integers(40, 150))
            al = int(rng.integers(0, q["a_right"] - m + 1))
            bl = int(rng.integers(1, 300))
            br = int(rng.integers(max(bl + 3 * m + 100, q["b_right"] - 500), q["b_right"] + 1))
            exg = (1, 1, 1, 1) if i % 3 else tuple(int(x) for x in rng.integers(0, 2, size=4))
            ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
                   fx["phs5"], fx["phs3"], al, al + m, bl, br, exg, exin=(q["b_left"], q["b_right"]))
        scores, cpos, ranges = eng.wip_udh_h(sc, ps, n_im)
        bad = []
        for i, p in enumerate(ps.items):
            ws, wcpos, wrng = oracle.wip_udh_h(sc, p, n_im)
            if int(scores[i]) != ws or ranges[i].tolist() != wrng.tolist() or cpos[i].tolist() != wcpos.tolist():
                bad.append((n_im, i, int(scores[i]), ws, ranges[i].tolist(), wrng.tolist(), cpos[i][:2].tolist(), wcpos[:2].tolist()))
        assert not bad, bad[:3]
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    for vmf in (100000, 30000):
        sc2 = spdg.scoring_h(fx, max_vmf_space=vmf)
        ps = abi.ProblemSetH()
        for i in range(10):
            m = int(rng.integers(80, q["a_right"] + 1))
            al = int(rng.integers(0, q["a_right"] - m + 1))
            ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
                   fx["phs5"], fx["phs3"], al, al + m, q["b_left"], q["b_right"], (1, 1, 1, 1),
                   exin=(q["b_left"], q["b_right"]), dinc=dinc)
        res = eng.align_h(sc2, ps)
        bad, n_ok = [], 0
        for i, (p, (score, skl, flag)) in enumerate(zip(ps.items, res)):
            try:
                ws, wskl = hh.align_h(sc2, p)
            except (hh.ReferenceUndefined, hh.NotRestated, hh.ReferenceFatal):
                assert flag != 0, i
                continue
            n_ok += 1
            if flag != 0 or score != ws or skl.ravel().tolist() != (wskl or []):
                bad.append((vmf, i, flag, score, ws, skl.ravel().tolist()[:12], (wskl or [])[:12]))
        assert n_ok >= 6 and not bad, bad[:3]
