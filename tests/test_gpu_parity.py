"""GPU parity (run with -m gpu on an MI355X): the HIP engines, called through the
C ABI, against (a) the committed reference goldens and (b) the CPU oracle on
the same inputs.  Integer work: every comparison is bit-exact.

Known, documented deviation (DESIGN.md "Local-mode corner"): none of the
all cases; hirschbergS1_wip with local ends runs in its own kernel (spdp_local_udh.hip).
"""
import re

import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files, golden_ids

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module", params=golden_files(), ids=golden_ids())
def fx(request):
    return spdg.load(request.param)


def _setup(fx, tag):
    sc = spdg.scoring(fx, nquant=(1 if tag == "q1" else None))
    ps, p = spdg.problem(fx)
    return sc, ps, p


def _row(r):
    from spaln_amd import abi
    r = [int(x) for x in r]
    if r[0] == abi.END_OF_ULK:
        return r[:1] + r[2:3]
    out = []
    for x in r:
        out.append(x)
        if x == abi.END_OF_ULK:
            break
    return out


@pytest.mark.parametrize("tag", ["qn", "q1"])
def test_scoreonly_vs_reference(eng, fx, tag):
    sc, ps, p = _setup(fx, tag)
    got = eng.wip_scoreonly(sc, ps)
    assert int(got[0]) == int(fx[f"wip_{tag}_score"][0])


@pytest.mark.parametrize("tag", ["qn", "q1"])
def test_forward_vs_reference(eng, fx, tag):
    sc, ps, p = _setup(fx, tag)
    (score, skl), = eng.wip_forward(sc, ps)
    assert score == int(fx[f"wip_{tag}_fwd_scr"][0])
    assert skl.ravel().tolist() == fx[f"wip_{tag}_fwd_skl"].tolist()


@pytest.mark.parametrize("tag", ["qn", "q1"])
def test_udh_vs_reference(eng, fx, tag):
    sc, ps, p = _setup(fx, tag)
    for k in [k for k in fx if re.fullmatch(rf"wip_{tag}_udh\d+_scr", k)]:
        n_im = int(re.search(r"udh(\d+)", k).group(1))
        scores, cpos, rng = eng.wip_udh(sc, ps, n_im)
        assert int(scores[0]) == int(fx[k][0]), k
        want = fx[f"wip_{tag}_udh{n_im}_cpos"].reshape(-1, 10)
        for i in range(n_im + 1):
            assert _row(cpos[0][i]) == _row(want[i]), (k, i)
        assert rng[0].tolist() == fx[f"wip_{tag}_udh{n_im}_rng"][:4].tolist(), k


def test_batch_vs_oracle(eng):
    """All goldens in ONE launch (different sizes, one queue) against the oracle."""
    from oracle import oracle
    from spaln_amd import abi
    groups = {}
    for f in golden_files():
        fx = spdg.load(f)
        if fx["prm"]["local"]:
            continue
        key = tuple(fx["qm_len"].tolist()) + (fx["prm"]["sh"],) + tuple(
            fx["prm"][k] for k in ("gop", "gep", "ipen", "llmt"))
        groups.setdefault(key, []).append(fx)
    for fxs in groups.values():
        sc = spdg.scoring(fxs[0])
        ps = abi.ProblemSet()
        for fx in fxs:
            spdg.problem(fx, ps)
        got = eng.wip_scoreonly(sc, ps)
        want = [oracle.wip_scoreonly(sc, p) for p in ps.items]
        assert got.tolist() == want


@pytest.mark.parametrize("alg", [2, 3])
def test_align_s_vs_reference(eng, fx, alg):
    """alignS_ng(ori=1, -Q0) through the C ABI: dispatch ladder + UDH + slab tracebacks + stdskl/trimskl."""
    sc = spdg.scoring(fx, nquant=(1 if alg == 3 else None))
    ps, p = spdg.problem(fx)
    (score, skl), = eng.align_s(sc, ps)
    assert score == int(fx[f"aln_scr_A{alg}"][0])
    assert skl.ravel().tolist() == fx[f"aln_skl_A{alg}"].tolist()


def test_many_problems_vs_oracle(eng):
    """A loaded GPU (hundreds of concurrent waves, ragged sizes): all three engines + the
    full alignS_ng ladder against the oracle, bit for bit.  Guards the in-wave
    store->load exchange of the boundary rows against ordering / caching hazards."""
    import numpy as np
    from oracle import oracle, host_logic
    from spaln_amd import abi, defaults, synth
    sc = defaults.scoring()
    rng = np.random.default_rng(4242)
    ps = abi.ProblemSet()
    for i in range(160):
        g = synth.make_gene(rng, n_exons=int(rng.integers(1, 7)), mrna_len=int(rng.integers(40, 900)),
                            flank=int(rng.integers(20, 400)), intron_hi=int(rng.integers(100, 2500)),
                            sub=float(rng.uniform(0, 0.1)), indel=float(rng.uniform(0, 0.02)))
        s5, s3 = synth.splice_signals(g.window)
        ps.add(defaults.encode(g.query), defaults.encode(g.window), s5, s3)
    got = eng.wip_scoreonly(sc, ps)
    want = [oracle.wip_scoreonly(sc, p) for p in ps.items]
    assert got.tolist() == want
    fwd = eng.wip_forward(sc, ps)
    for (s, skl), p in zip(fwd, ps.items):
        ws, wskl = oracle.wip_forward(sc, p)
        assert s == ws and np.array_equal(skl, wskl)
    us, ucpos, urng = eng.wip_udh(sc, ps, 4)
    for i, p in enumerate(ps.items):
        ws, wcpos, wrng = oracle.wip_udh(sc, p, 4)
        assert int(us[i]) == ws and urng[i].tolist() == wrng.tolist()
        for r in range(5):
            assert _row(ucpos[i][r]) == _row(wcpos[r])
    sc2 = defaults.scoring(max_vmf_space=400000)          # forces UDH + slabs on these sizes
    al = eng.align_s(sc2, ps)
    for (s, skl), p in zip(al, ps.items):
        ws, wskl = host_logic.align_s(sc2, p)
        assert s == ws and skl.ravel().tolist() == (wskl or [])


def test_scalar_engines_vs_reference(eng, fx):
    """Scalar exact-ILD engines on the GPU (-A0: scorealoneS_ng, forwardS_ng) vs the reference."""
    from oracle import oracle, host_logic
    from tests.test_oracle_scalar import direct_under_a0
    sc = spdg.scoring(fx)
    ps, p = spdg.problem(fx)
    assert int(eng.scalar_scorealone(sc, ps)[0]) == int(fx["hom_scr_A0"][0])
    (scr, skl), = eng.scalar_forward(sc, ps)
    oscr, oskl = oracle.scalar_forward(sc, p)
    assert scr == oscr and skl.tolist() == oskl.tolist()
    if direct_under_a0(fx, p, oracle.stripe(p, sc.sh)):
        assert scr == int(fx["aln_scr_A0"][0])
        rec = [(int(a), int(b)) for a, b in skl]
        fin = host_logic.trim_skl(host_logic.std_skl(rec), p) if len(rec) >= 2 else []
        flat = ([1, len(fin)] + [x for mn in fin for x in mn]) if fin else []
        assert flat == fx["aln_skl_A0"].tolist()


def test_skl_rng_s_goldens(eng):
    """spdp_skl_rng_s (skl_rngS_ng on the device) on the reference's own corner lists: total score,
    alignment statistics and every per-exon record, for all four engine selectors' alignments"""
    bad = []
    n_checked = 0
    for f in golden_files():
        fx = spdg.load(f)
        for alg in (0, 1, 2, 3):
            if f"rng_eij_A{alg}" not in fx:
                continue
            sc = spdg.scoring(fx, nquant=(1 if alg == 3 else None))
            ps, p = spdg.problem(fx)
            fs = fx[f"rng_fstat_A{alg}"]
            (score, fst, ex), = eng.skl_rng_s(sc, ps, [fx[f"aln_skl_A{alg}"].reshape(-1, 2)],
                                               codonk1=fx["prm"]["codonk1"], minl=fx["prm"]["minl"],
                                               jneibr=int(fs[6]), lsg=int(fs[7]))
            ok = (score == int(fx[f"rng_scr_A{alg}"][0]) and fst == [int(x) for x in fs[:5]]
                  and ex.tolist() == fx[f"rng_eij_A{alg}"].reshape(-1, 21).tolist())
            n_checked += 1
            if not ok:
                bad.append((f.split("/")[-1], alg, score, int(fx[f"rng_scr_A{alg}"][0]), fst, fs[:5].tolist()))
    assert n_checked >= 100 and not bad, bad[:4]


def test_align_then_rescore_batch(eng):
    """the product pipeline: spdp_align_s then spdp_skl_rng_s on its output, a whole batch per call"""
    from spaln_amd import abi
    fxs = [spdg.load(f) for f in golden_files() if "local" not in f and "tiny" not in f and "exg" not in f
           and "narrow" not in f]
    fxs = [fx for fx in fxs if "rng_eij_A2" in fx and fx["prm"]["sh"] == 100]
    sc = spdg.scoring(max(fxs, key=lambda fx: len(fx["intpen"])))     # the table must cover the longest window
    ps = abi.ProblemSet()
    for fx in fxs:
        spdg.problem(fx, ps)
    aln = eng.align_s(sc, ps)
    fs = fxs[0]["rng_fstat_A2"]
    res = eng.skl_rng_s(sc, ps, [skl for _, skl in aln], codonk1=fxs[0]["prm"]["codonk1"],
                        minl=fxs[0]["prm"]["minl"], jneibr=int(fs[6]), lsg=int(fs[7]))
    assert len(res) == len(fxs) >= 8
    # the fixtures were taken under different MaxVmfSpace / ubh settings, so the reference to compare
    # with under ONE scoring bundle is the oracle pipeline (ladder + rescoring restatements)
    from oracle import host_logic
    for p, (score, fst, ex) in zip(ps.items, res):
        _, wskl = host_logic.align_s(sc, p)
        wh, wfst, wrecs = host_logic.skl_rng_s(sc, p, wskl, codonk1=fxs[0]["prm"]["codonk1"],
                                               minl=fxs[0]["prm"]["minl"], jneibr=int(fs[6]), lsg=int(fs[7]))
        assert score == wh and fst == wfst and ex.tolist() == wrecs


def test_homscore_s_ng_goldens(eng):
    from spaln_amd import abi
    """spdp_homscore_s = HomScoreS_ng (-A2 / -A3 / -A1), every fixture in one batch per intron model: the
    `_wip` engine, scoreonlyS1 under -A1 (SpdpScoring.scalar_engines = 2), and scorealoneS_ng for the query
    ranges below 4 rows"""
    from tests.conftest import golden_files
    for alg in (2, 3, 1):
        cases = [spdg.load(f) for f in golden_files("s1_") if "local" not in f]
        ref = max(cases, key=lambda fx: fx["intpen"].size)
        for sh in sorted({fx["prm"]["sh"] for fx in cases}):
            sub = [fx for fx in cases if fx["prm"]["sh"] == sh]
            sc = spdg.scoring(ref, nquant=1 if alg == 3 else None, sh=sh, scalar_engines=2 if alg == 1 else 0)
            ps = abi.ProblemSet()
            for fx in sub:
                spdg.problem(fx, ps)
            got = eng.homscore_s(sc, ps)
            assert got.tolist() == [int(fx[f"hom_scr_A{alg}"][0]) for fx in sub], alg


def test_align_s_ori3_against_oracle(eng):
    from spaln_amd import abi
    """alignS_ng(ori = 3), -Q0: infer_orientation + one alignment; the 'reverse' problems here are other
    fixtures' windows, so that either orientation wins for some queries"""
    from oracle import host_logic
    from tests.conftest import golden_files
    names = ["s1_basic", "s1_single_exon", "s1_divergent", "s1_random", "s1_tiny_m3", "s1_cut_left"]
    fxs = [spdg.load([f for f in golden_files("s1_") if f.endswith(n + ".spdg")][0]) for n in names]
    sc = spdg.scoring(max(fxs, key=lambda fx: fx["intpen"].size))
    fwd, rev = abi.ProblemSet(), abi.ProblemSet()
    for i, fx in enumerate(fxs):
        spdg.problem(fx, fwd)
        spdg.problem(fxs[(i + 1) % len(fxs)] if i % 2 else fxs[i], rev)      # odd: another locus, even: identical
    res, orient = eng.align_s_ori3(sc, fwd, rev)
    picked = []
    for i, (pf, pr) in enumerate(zip(fwd.items, rev.items)):
        (ws, wskl), wori = host_logic.align_s_ori3(sc, pf, pr)
        assert int(orient[i]) == wori
        assert res[i][0] == ws and res[i][1].ravel().tolist() == (wskl or [])
        picked.append(wori)
    assert picked[0] == 0 and picked[2] == 0                   # ties keep the query as given


def test_homscore_a1_local_and_subranges(eng):
    """scoreonlyS1 in local mode (goldens) and on random sub-ranges / end-gap flags against the oracle"""
    from spaln_amd import abi, synth
    from oracle import oracle
    from tests.conftest import golden_files
    for name in ("s1_local", "s1_local_cut"):
        fx = spdg.load([f for f in golden_files("s1_") if f.endswith(name + ".spdg")][0])
        sc = spdg.scoring(fx, scalar_engines=2)
        ps, _ = spdg.problem(fx)
        assert int(eng.homscore_s(sc, ps)[0]) == int(fx["hom_scr_A1"][0])
    fx = spdg.load([f for f in golden_files("s1_") if f.endswith("s1_indels.spdg")][0])
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 97)
    extra = dict(cano5=fx["cano5"], cano3=fx["cano3"],
                 dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
    for local in (0, 1):
        sc = spdg.scoring(fx, scalar_engines=2, local=local)
        ps = abi.ProblemSet()
        for i in range(96):
            al = int(rng.integers(0, q["a_right"] - 40))
            ar = int(rng.integers(al + 4, min(al + 400, q["a_right"]) + 1))
            bl = int(rng.integers(0, q["b_right"] - 600))
            br = int(rng.integers(bl + (ar - al) // 2 + 20, min(bl + 3000, q["b_right"]) + 1))
            exg = tuple(int(x) for x in rng.integers(0, 2, size=4))
            ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, ar, bl, br, exg, **extra)
        got = eng.homscore_s(sc, ps)
        want = [oracle.exact_scoreonly(sc, p) for p in ps.items]
        bad = [(i, int(g), w) for i, (g, w) in enumerate(zip(got, want)) if int(g) != w]
        assert not bad, (local, bad[:5])


def test_align_a1_goldens(eng):
    """alignS_ng under -A1 (SpdpScoring.scalar_engines = 2): forwardS1 / hirschbergS1 on the GPU + the ladder,
    against the reference's -A1 alignments (all fixtures: traceback and linear-space branches)"""
    from spaln_amd import abi
    from oracle import host_logic
    from tests.conftest import golden_files
    n_ok = n_ls = 0
    for local in (False, True):
        cases = [spdg.load(f) for f in golden_files("s1_") if ("local" in f) == local]
        ref = max(cases, key=lambda fx: fx["intpen"].size)
        key = lambda fx: (fx["prm"]["max_vmf_space"], fx["prm"]["ubh"], fx["prm"]["sh"])
        for vmf, ubh, sh in sorted({key(fx) for fx in cases}):
            sub = [fx for fx in cases if key(fx) == (vmf, ubh, sh)]
            sc = spdg.scoring(ref, scalar_engines=2, max_vmf_space=vmf, ubh=ubh, sh=sh, local=1 if local else 0)
            ps = abi.ProblemSet()
            for fx in sub:
                spdg.problem(fx, ps)
            res = eng.align_s(sc, ps, allow_partial=True)
            for fx, p, (score, skl) in zip(sub, ps.items, res):
                try:
                    host_logic.align_s(sc, p, simd=1)
                except host_logic.NeedsScalarEngine:
                    assert len(skl) == 0
                    n_ls += 1
                    continue
                assert score == int(fx["aln_scr_A1"][0])
                assert skl.ravel().tolist() == fx["aln_skl_A1"].tolist()
                n_ok += 1
    assert n_ok == len(golden_files("s1_")) >= 34 and n_ls == 0


def test_forward_s1_subranges_against_oracle(eng):
    """forwardS1 through the -A1 ladder on random sub-ranges / end-gap flags, global and local"""
    from spaln_amd import abi, synth
    from oracle import host_logic
    from tests.conftest import golden_files
    fx = spdg.load([f for f in golden_files("s1_") if f.endswith("s1_basic.spdg")][0])
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 98)
    extra = dict(cano5=fx["cano5"], cano3=fx["cano3"],
                 dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
    for local in (0, 1):
        sc = spdg.scoring(fx, scalar_engines=2, local=local)
        ps = abi.ProblemSet()
        for i in range(64):
            al = int(rng.integers(0, q["a_right"] - 40))
            ar = int(rng.integers(al + 8, min(al + 300, q["a_right"]) + 1))
            bl = int(rng.integers(0, q["b_right"] - 500))
            br = int(rng.integers(bl + (ar - al) // 2 + 20, min(bl + 2500, q["b_right"]) + 1))
            exg = tuple(int(x) for x in rng.integers(0, 2, size=4))
            ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, ar, bl, br, exg, **extra)
        res = eng.align_s(sc, ps, allow_partial=True)
        bad, n_ok = [], 0
        for i, (p, (score, skl)) in enumerate(zip(ps.items, res)):
            try:
                ws, wskl = host_logic.align_s(sc, p, simd=1)
            except (host_logic.NeedsScalarEngine, host_logic.ReferenceUndefined):
                continue
            n_ok += 1
            if score != ws or skl.ravel().tolist() != (wskl or []):
                bad.append((local, i, (p.a_left, p.a_right, p.b_left, p.b_right), score, ws,
                            skl.ravel().tolist()[:12], (wskl or [])[:12]))
        assert n_ok >= 40 and not bad, bad[:3]


def test_a1_ladder_linear_space_against_oracle(eng):
    """alignS_ng under -A1 pushed into the linear-space branches (small MaxVmfSpace: recurrent and recursive)
    on sub-ranges, vs the oracle ladder (hirschbergS1 + forwardS1 per slab)"""
    from spaln_amd import abi, synth
    from oracle import host_logic
    from tests.conftest import golden_files
    fx = spdg.load([f for f in golden_files("s1_") if f.endswith("s1_auto_udh.spdg")][0])
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 99)
    extra = dict(cano5=fx["cano5"], cano3=fx["cano3"],
                 dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
    for vmf, ubh in ((600000, 0), (100000, 3), (40000, 0)):
        sc = spdg.scoring(fx, scalar_engines=2, max_vmf_space=vmf, ubh=ubh)
        ps = abi.ProblemSet()
        for i in range(16):
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
                ws, wskl = host_logic.align_s(sc, p, simd=1)
            except (host_logic.NeedsScalarEngine, host_logic.ReferenceUndefined):
                continue
            n_ok += 1
            if score != ws or skl.ravel().tolist() != (wskl or []):
                bad.append((vmf, i, (p.a_left, p.a_right, p.b_left, p.b_right), score, ws,
                            skl.ravel().tolist()[:12], (wskl or [])[:12]))
        assert n_ok >= 12 and not bad, bad[:3]


def test_align_a6_recursive_switch(eng):
    """-A6 (algmode.alg & 4 with the `_wip` engines): SpdpScoring.recursive sends lspS_ng down the recursive
    linear-space branch; against the reference's -A6 output (flat penalty model, see the CPU test)"""
    from spaln_amd import abi
    from tests.conftest import golden_files
    for local in (False, True):
        cases = [spdg.load(f) for f in golden_files("s1_") if ("local" in f) == local]
        ref = max(cases, key=lambda fx: fx["intpen"].size)
        key = lambda fx: (fx["prm"]["max_vmf_space"], fx["prm"]["ubh"], fx["prm"]["sh"])
        for vmf, ubh, sh in sorted({key(fx) for fx in cases}):
            sub = [fx for fx in cases if key(fx) == (vmf, ubh, sh)]
            sc = spdg.scoring(ref, nquant=1, recursive=1, max_vmf_space=vmf, ubh=ubh, sh=sh, local=1 if local else 0)
            ps = abi.ProblemSet()
            for fx in sub:
                spdg.problem(fx, ps)
            res = eng.align_s(sc, ps)
            for fx, (score, skl) in zip(sub, res):
                assert score == int(fx["aln_scr_A6"][0])
                assert skl.ravel().tolist() == fx["aln_skl_A6"].tolist()


def test_local_udh_against_oracle(eng):
    """hirschbergS1_wip with local ends (-LS, spdp_local_udh.hip) on random sub-ranges: cpos rows, ranges and
    score against the oracle (which reproduces the reference's stale lanes, pinned by s1_local_cut), and
    alignS_ng pushed into the linear-space branches in local mode"""
    from spaln_amd import abi, synth
    from oracle import oracle, host_logic
    from tests.conftest import golden_files
    fx = spdg.load([f for f in golden_files("s1_") if f.endswith("s1_local.spdg")][0])
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + 98)
    sc = spdg.scoring(fx)
    assert sc.local
    for n_im in (1, 3):
        ps = abi.ProblemSet()
        for i in range(24):
            m = int(rng.integers(100, 400))
            al = int(rng.integers(0, q["a_right"] - m))
            bl = int(rng.integers(0, 300))
            br = int(rng.integers(max(bl + m + 200, q["b_right"] - 600), q["b_right"] + 1))
            exg = (1, 1, 1, 1) if i % 3 else tuple(int(x) for x in rng.integers(0, 2, size=4))
            ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, al + m, bl, br, exg)
        scores, cpos, rngs = eng.wip_udh(sc, ps, n_im)
        bad = []
        for i, p in enumerate(ps.items):
            ws, wcpos, wrng = oracle.wip_udh(sc, p, n_im)
            if int(scores[i]) != ws or rngs[i].tolist() != wrng.tolist() or \
                    [_row(x) for x in cpos[i]] != [_row(x) for x in wcpos]:
                bad.append((n_im, i, int(scores[i]), ws, rngs[i].tolist(), wrng.tolist(), cpos[i][:2].tolist(), wcpos[:2].tolist()))
        assert not bad, bad[:3]
    for vmf in (400000, 100000):
        sc2 = spdg.scoring(fx, max_vmf_space=vmf)
        ps = abi.ProblemSet()
        for i in range(12):
            m = int(rng.integers(250, q["a_right"]))
            al = int(rng.integers(0, q["a_right"] - m + 1))
            ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, al + m, 0, q["b_right"], (1, 1, 1, 1))
        res = eng.align_s(sc2, ps)
        bad = []
        for i, p in enumerate(ps.items):
            ws, wskl = host_logic.align_s(sc2, p)
            if res[i][0] != ws or res[i][1].ravel().tolist() != (wskl or []):
                bad.append((vmf, i, res[i][0], ws, res[i][1].ravel().tolist()[:12], (wskl or [])[:12]))
        assert not bad, bad[:3]


def test_skl_edits_s_goldens(eng):
    """spdp_skl_edits_s: the edit records skl_rngS_ng collects for its Cigar / Vulgar / SAM writers, on the reference's
    own corner lists (-A2 and -A0 alignments of every fixture), record for record; SAM header fields included"""
    from spaln_amd import abi
    bad, n_checked, n_introns = [], 0, 0
    for f in golden_files() + golden_files("c2_") + golden_files("o3_"):
        fx = spdg.load(f)
        for alg in (0, 2):
            if f"rng_cigar_A{alg}" not in fx:
                continue
            sc = spdg.scoring(fx)
            ps, p = spdg.problem(fx)
            fs = fx[f"rng_fstat_A{alg}"]
            kw = dict(codonk1=fx["prm"]["codonk1"], minl=fx["prm"]["minl"], jneibr=int(fs[6]), lsg=int(fs[7]))
            skl = [fx[f"aln_skl_A{alg}"].reshape(-1, 2)]
            (cig, _), = eng.skl_edits_s(sc, ps, skl, abi.FMT_CIGAR, **kw)
            (vul, _), = eng.skl_edits_s(sc, ps, skl, abi.FMT_VULGAR, **kw)
            (sam, hdr), = eng.skl_edits_s(sc, ps, skl, abi.FMT_SAM, **kw)
            ok = (cig[:, :2].ravel().tolist() == fx[f"rng_cigar_A{alg}"].tolist()
                  and vul.ravel().tolist() == fx[f"rng_vulgar_A{alg}"].tolist()
                  and sam[:, :2].ravel().tolist() == fx[f"rng_sam_A{alg}"].tolist()
                  and hdr == fx[f"rng_samhdr_A{alg}"].tolist())
            n_checked += 1
            n_introns += int((cig[:, 0] == ord("N")).sum())
            if not ok:
                bad.append((f.split("/")[-1], alg, cig[:6].tolist(), fx[f"rng_cigar_A{alg}"][:12].tolist(), hdr,
                            fx[f"rng_samhdr_A{alg}"].tolist()))
    assert n_checked >= 60 and n_introns >= 100 and not bad, bad[:3]


def test_exon_form_from_device_records(eng):
    """the reporting pipeline end to end: spdp_skl_rng_s on the device, then spdp_exon_form_text on its records, equals the
    reference's -O4 lines for its own alignment (tests/test_exon_form.py does the same from the reference's records)"""
    from spaln_amd import engine as _engine
    from tests.test_exon_form import _text      # noqa: F401  (same parameter decoding)
    n = 0
    for f in golden_files() + golden_files("c2_"):
        fx = spdg.load(f)
        for alg in (0, 2):
            if f"rng_exn_A{alg}" not in fx:
                continue
            sc = spdg.scoring(fx)
            ps, p = spdg.problem(fx)
            fs = fx[f"rng_fstat_A{alg}"]
            (score, fst, ex), = eng.skl_rng_s(sc, ps, [fx[f"aln_skl_A{alg}"].reshape(-1, 2)],
                                               codonk1=fx["prm"]["codonk1"], minl=fx["prm"]["minl"],
                                               jneibr=int(fs[6]), lsg=int(fs[7]))
            fx2 = dict(fx)
            fx2[f"rng_eij_A{alg}"] = ex
            want = bytes(fx[f"rng_exn_A{alg}"])
            _, _, got = _text(fx2, alg, False, eng.lib, header=want.startswith(b"#"))
            assert got == want, (f, alg)
            n += 1
    assert n >= 60
