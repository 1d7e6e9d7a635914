#!/usr/bin/env python3
"""Larger randomised runs of the exactness kernels against the oracle than the test suite affords
(GPU box: python tools/fuzz_exact.py [n [seed]]).  Protein -A1 forward / linear-space engines and the local-ends
linear-space kernels of both paths on random sub-ranges with random end-gap flags.  The last section runs whole -A0 / -A1
ladders; a mismatch there is looked into with tools/ladder_case.py (the four engine / MaxVmfSpace combinations),
tools/ladder_subproblems.py (which linear-space call differs) and tools/a1_udh_case.py (that call's cpos rows): so far every
one was a path down the window's left edge or an empty optimum, where the reference's own link walk reads stale lanes
(2 of 2400 ladders with seed 99173)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi, engine, synth
from oracle import oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
eng = engine.Engine(0)
bad = 0


def h_sub(fx, n, m_lo, m_hi, with_dinc=True):
    q = fx["prm"]
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    ps = abi.ProblemSetH()
    for i in range(n):
        hi = min(m_hi, q["a_right"])
        m = int(rng.integers(min(m_lo, hi), hi + 1))
        al = int(rng.integers(0, q["a_right"] - m + 1))
        bl = int(rng.integers(1, 600))
        br = int(rng.integers(min(q["b_right"], max(bl + 3 * m + 60, q["b_right"] - 2500)), q["b_right"] + 1))
        exg = (1, 1, 1, 1) if i % 3 == 0 else tuple(int(x) for x in rng.integers(0, 3, size=4) % 2)
        kw = dict(dinc=dinc) if with_dinc else {}
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
               fx["phs5"], fx["phs3"], al, al + m, bl, br, exg, exin=(q["b_left"], q["b_right"]), **kw)
    return ps


H = {os.path.basename(f)[:-5]: f for f in golden_files("h1_")}
S = {os.path.basename(f)[:-5]: f for f in golden_files("s1_")}
for name, local in (("h1_400aa", 0), ("h1_local", 1)):
    fx = spdg.load(H[name])
    sc = spdg.scoring_h(fx, scalar_engines=2)
    ps = h_sub(fx, N, 8, 200)
    res = eng.scalar_forward_h(sc, ps)
    for i, (p, (score, skl)) in enumerate(zip(ps.items, res)):
        ws, wskl, wflag = oracle.exact_forward_h(sc, p)
        if not wflag and (score != ws or skl.ravel().tolist() != wskl.ravel().tolist()):
            bad += 1; print("forwardH1", name, i, (p.a_left, p.a_right, p.b_left, p.b_right), score, ws)
    for n_im in (1, 2, 4):
        ps = h_sub(fx, N // 4, 40 * (n_im + 1), 60 * (n_im + 1))
        scores, cpos, ranges, flags = eng.scalar_udh_h(sc, ps, n_im, 1)
        for i, p in enumerate(ps.items):
            ws, wc, wr = oracle.exact_udh_h(sc, p, n_im)
            if int(scores[i]) != ws or ranges[i].tolist() != wr.tolist() or cpos[i].tolist() != wc.tolist():
                bad += 1; print("hirschbergH1", name, n_im, i, (p.a_left, p.a_right, p.b_left, p.b_right), int(scores[i]), ws)
    if local:
        scw = spdg.scoring_h(fx)
        for n_im in (1, 2, 4):
            ps = h_sub(fx, N // 4, 30 * (n_im + 1), 35 * (n_im + 1), with_dinc=False)
            scores, cpos, ranges = eng.wip_udh_h(scw, ps, n_im)
            for i, p in enumerate(ps.items):
                ws, wc, wr = oracle.wip_udh_h(scw, p, n_im)
                if int(scores[i]) != ws or ranges[i].tolist() != wr.tolist() or cpos[i].tolist() != wc.tolist():
                    bad += 1; print("hirschbergH1_wip -LS", n_im, i, (p.a_left, p.a_right, p.b_left, p.b_right), int(scores[i]), ws)

fx = spdg.load(S["s1_local"])
q = fx["prm"]
sc = spdg.scoring(fx)
for n_im in (1, 2, 5):
    ps = abi.ProblemSet()
    for i in range(N // 2):
        m = int(rng.integers(min(40 * (n_im + 1), q["a_right"] - 1), q["a_right"]))
        al = int(rng.integers(0, q["a_right"] - m + 1))
        bl = int(rng.integers(0, 300))
        br = int(rng.integers(max(bl + m + 100, q["b_right"] - 600), q["b_right"] + 1))
        exg = (1, 1, 1, 1) if i % 3 == 0 else tuple(int(x) for x in rng.integers(0, 2, size=4))
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, al + m, bl, br, exg)
    scores, cpos, rngs = eng.wip_udh(sc, ps, n_im)
    for i, p in enumerate(ps.items):
        ws, wc, wr = oracle.wip_udh(sc, p, n_im)
        same = [[int(x) for x in r] for r in cpos[i]] == [[int(x) for x in r] for r in wc]
        if int(scores[i]) != ws or rngs[i].tolist() != wr.tolist() or (not same and wc[0][0] != abi.END_OF_ULK):
            bad += 1; print("hirschbergS1_wip -LS", n_im, i, (p.a_left, p.a_right, p.b_left, p.b_right), int(scores[i]), ws)
# alignS_ng under -A1 with local ends pushed into the linear-space branches (hirschbergS1 -LS through the ladder)
from oracle import host_logic
extra = dict(cano5=fx["cano5"], cano3=fx["cano3"], dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
for vmf in (300000, 120000):
    sc1 = spdg.scoring(fx, scalar_engines=2, max_vmf_space=vmf)
    ps = abi.ProblemSet()
    for i in range(max(8, N // 8)):
        m = int(rng.integers(200, q["a_right"] + 1))
        al = int(rng.integers(0, q["a_right"] - m + 1))
        bl = int(rng.integers(0, 200))
        br = int(rng.integers(q["b_right"] - 300, q["b_right"] + 1))
        exg = (1, 1, 1, 1) if i % 2 == 0 else tuple(int(x) for x in rng.integers(0, 2, size=4))
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, al + m, bl, br, exg, **extra)
    res = eng.align_s(sc1, ps, allow_partial=True)
    for i, (p, (score, skl)) in enumerate(zip(ps.items, res)):
        try:
            ws, wskl = host_logic.align_s(sc1, p, simd=1)
        except (host_logic.NeedsScalarEngine, host_logic.ReferenceUndefined):
            continue
        if score != ws or skl.ravel().tolist() != (wskl or []):
            bad += 1; print("alignS_ng -A1 -LS", vmf, i, (p.a_left, p.a_right, p.b_left, p.b_right), score, ws)
# alignS_ng under -A0 and -A1 on a global fixture, tall sub-ranges (the tiles / stripes of a problem run as pipelines
# of waves), traceback and linear-space branches
fx = spdg.load(S["s1_1400nt"])
q = fx["prm"]
extra = dict(cano5=fx["cano5"], cano3=fx["cano3"], dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
for eng_sel, simd in ((1, 0), (2, 1)):
    for vmf in (1 << 30, 600000, 150000):
        sc1 = spdg.scoring(fx, scalar_engines=eng_sel, max_vmf_space=vmf)
        ps = abi.ProblemSet()
        for i in range(max(8, N // 8)):
            m = int(rng.integers(100, q["a_right"] + 1))
            al = int(rng.integers(0, q["a_right"] - m + 1))
            bl = int(rng.integers(0, 600))
            br = int(rng.integers(max(bl + m + 300, q["b_right"] - 1500), q["b_right"] + 1))
            exg = (1, 1, 1, 1) if i % 2 == 0 else tuple(int(x) for x in rng.integers(0, 2, size=4))
            ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, al + m, bl, br, exg, **extra)
        res = eng.align_s(sc1, ps, allow_partial=True)
        hom = eng.homscore_s(sc1, ps)
        n_cmp = 0
        for i, (p, (score, skl)) in enumerate(zip(ps.items, res)):
            try:
                ws, wskl = host_logic.align_s(sc1, p, simd=simd)
            except (host_logic.NeedsScalarEngine, host_logic.ReferenceUndefined):
                continue
            n_cmp += 1
            if len(skl) == 0 and not wskl and min(score, ws) <= abi.NEVSEL and p.a_exgl and p.a_exgr and p.b_exgl and p.b_exgr:
                continue        # empty optimum with every end free: neither side aligns anything; the link walk's verdict is stale-lane arithmetic
            if score != ws or skl.ravel().tolist() != (wskl or []):
                bad += 1; print("alignS_ng", "-A0" if simd == 0 else "-A1", vmf, i, (p.a_left, p.a_right, p.b_left, p.b_right),
                                "exg", (p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr), score, ws, skl.ravel().tolist()[:12], (wskl or [])[:12])
        print("alignS_ng", "-A0" if simd == 0 else "-A1", "MaxVmfSpace", vmf, ":", n_cmp, "compared", flush=True)
eng.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
