"""Randomised parity: GPU engines vs the oracle on small problems whose EVERYTHING is random --
sequences, signals, gap / intron parameters, quantile tables, band shoulder, end-gap flags,
sub-ranges.  The goldens pin the oracle to the reference under its default parameters; this keeps
the kernels honest everywhere else in the input space."""
import numpy as np
import pytest

from spaln_amd import abi, defaults, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


def _on_path(skl, m, n, step):
    """is cell (m, n) on the polyline through the corner records (any order)?"""
    pts = sorted((int(a), int(b)) for a, b in skl)
    for (m0, n0), (m1, n1) in zip(pts[:-1], pts[1:]):
        if not (m0 <= m <= m1 and n0 <= n <= n1):
            continue
        if m0 == m1 or n0 == n1:
            return True
        if n - n0 == step * (m - m0) or n1 - n == step * (m1 - m):
            return True
    return False


def _well_defined(wrng, wcpos, fskl, step):
    """The linear-space result is the reference's own only when its links stayed on computed cells:
    the written-back range equals the traceback engine's end points and every intermediate-row
    crossing lies on that engine's path.  Otherwise the path ran along the free left edge through
    lanes that have not reached the window -- their links are whatever the previous stripe left in
    hc_a / fc_a (never re-initialised) -- and the two reference engines
    disagree with each other."""
    if wrng[0] >= wrng[1] or wrng[2] >= wrng[3] or len(fskl) < 2:
        return False
    pts = sorted((int(a), int(b)) for a, b in fskl)
    if pts[0] != (wrng[0], wrng[2]) or pts[-1] != (wrng[1], wrng[3]):
        return False
    for row in wcpos:
        if row[0] < abi.END_OF_ULK and not _on_path(fskl, int(row[0]), int(row[2]), step):
            return False
    return True


def _rand_scoring_s(rng, local=0):
    nq = int(rng.integers(1, 6))
    qlen = np.sort(rng.choice(np.arange(30, 1500), size=5, replace=False))
    qpen = -rng.integers(150, 400, size=5)
    return defaults.scoring(gop=-int(rng.integers(20, 120)), gep=-int(rng.integers(5, 40)),
                            ipen=-int(rng.integers(100, 400)), llmt=int(rng.integers(5, 40)),
                            qm_len=[int(x) for x in qlen], qm_pen=[int(x) for x in qpen], nquant=nq,
                            sh=int(rng.choice([10, 30, 100])), local=local)


def _rand_problem_s(rng, ps):
    m = int(rng.integers(9, 120))
    n = int(rng.integers(m + 20, 900))
    g = synth.make_gene(rng, n_exons=int(rng.integers(1, 4)), mrna_len=max(m, 40), flank=int(rng.integers(10, 80)),
                        intron_hi=int(rng.integers(80, 400)), sub=float(rng.uniform(0, 0.3)))
    w, q = defaults.encode(g.window), defaults.encode(g.query)
    N = w.size + 1
    s5 = rng.integers(-900, 150, size=N).astype(np.int16)
    s3 = rng.integers(-900, 150, size=N).astype(np.int16)
    al = int(rng.integers(0, max(1, q.size // 4)))
    ar = int(rng.integers(max(al + 9, q.size // 2), q.size + 1))
    bl = int(rng.integers(0, max(1, w.size // 5)))
    br = int(rng.integers(max(bl + (ar - al) + 5, w.size // 2), w.size + 1))
    exg = tuple(int(x) for x in rng.integers(0, 2, size=4))
    return ps.add(q, w, s5, s3, al, ar, bl, br, exg)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_cdna_engines(eng, seed):
    from oracle import oracle
    rng = np.random.default_rng(synth.SEED + 9000 + seed)
    n_empty = n_full = 0
    for rnd in range(4):
        sc = _rand_scoring_s(rng)
        ps = abi.ProblemSet()
        for _ in range(40):
            _rand_problem_s(rng, ps)
        got = eng.wip_scoreonly(sc, ps)
        assert got.tolist() == [oracle.wip_scoreonly(sc, p) for p in ps.items]
        for (s, skl), p in zip(eng.wip_forward(sc, ps), ps.items):
            ws, wskl = oracle.wip_forward(sc, p)
            assert s == ws and skl.tolist() == wskl.tolist()
        big = abi.ProblemSet()
        for p in ps.items:
            if p.a_right - p.a_left >= 40:
                big.items.append(p)
        big._keep = ps._keep
        if rnd == 3:
            assert n_full > n_empty
        if len(big):
            n_im = int(rng.integers(1, 3))
            us, ucpos, urng = eng.wip_udh(sc, big, n_im)
            for i, p in enumerate(big.items):
                ws, wcpos, wrng = oracle.wip_udh(sc, p, n_im)
                if not _well_defined(wrng, wcpos, oracle.wip_forward(sc, p)[1], 1):
                    n_empty += 1
                    continue
                assert int(us[i]) == ws and urng[i].tolist() == wrng.tolist()
                assert ucpos[i].tolist() == wcpos.tolist()
                n_full += 1


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_cdna_local(eng, seed):
    """-LS: Kadane reset at the left end, running maximum at the right end (score-only and traceback)"""
    from oracle import oracle
    rng = np.random.default_rng(synth.SEED + 9050 + seed)
    for rnd in range(3):
        sc = _rand_scoring_s(rng, local=1)
        ps = abi.ProblemSet()
        for _ in range(40):
            _rand_problem_s(rng, ps)
        assert eng.wip_scoreonly(sc, ps).tolist() == [oracle.wip_scoreonly(sc, p) for p in ps.items]
        for (s, skl), p in zip(eng.wip_forward(sc, ps), ps.items):
            ws, wskl = oracle.wip_forward(sc, p)
            assert s == ws and skl.tolist() == wskl.tolist()


def _rand_scoring_h(rng, local=0):
    nq = int(rng.integers(1, 6))
    qlen = np.sort(rng.choice(np.arange(30, 1500), size=5, replace=False))
    qpen = -rng.integers(250, 600, size=5)
    return defaults.scoring_h(gop=-int(rng.integers(40, 150)), gep=-int(rng.integers(8, 40)),
                              gapw1=-int(rng.integers(200, 500)), gapw2=-int(rng.integers(200, 500)),
                              gapw3=-int(rng.integers(60, 200)), ipen=-int(rng.integers(200, 500)),
                              llmt=int(rng.integers(5, 40)), qm_len=[int(x) for x in qlen],
                              qm_pen=[int(x) for x in qpen], nquant=nq, sh=int(rng.choice([5, 20, 100])),
                              term_codon=int(rng.integers(0, 2)), local=local)


def _rand_problem_h(rng, ps):
    aa = int(rng.integers(10, 90))
    g = synth.make_protein_gene(rng, n_exons=int(rng.integers(1, 4)), aa_len=aa, flank=int(rng.integers(10, 120)),
                                sub=float(rng.uniform(0, 0.4)), intron_hi=int(rng.integers(80, 400)))
    sg = synth.protein_signals(g.window, rng)
    q = synth.encode_protein(g.query)
    L = g.window.size
    al = int(rng.integers(0, max(1, q.size // 4)))
    ar = int(rng.integers(max(al + 9, q.size // 2), q.size + 1))
    bl = int(rng.integers(0, max(1, L // 5)))
    br = int(rng.integers(max(bl + 3 * (ar - al) // 2, L // 2), L + 1))
    exg = tuple(int(x) for x in rng.integers(0, 2, size=4))
    return ps.add(q, sg["b"], sg["sig5"], sg["sig3"], sg["sigS"], sg["sigT"], sg["sigE"], sg["phs5"], sg["phs3"],
                  al, ar, bl, br, exg, exin=(0, L))


@pytest.mark.parametrize("local", [0, 1])
@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_protein_forward(eng, seed, local):
    from oracle import oracle
    rng = np.random.default_rng(synth.SEED + 9100 + seed + 10 * local)
    flagmap = {0: 0, -2: -1, -3: -2}
    for rnd in range(4):
        sc = _rand_scoring_h(rng, local)
        ps = abi.ProblemSetH()
        for _ in range(40):
            _rand_problem_h(rng, ps)
        for i, ((s, skl, flag), p) in enumerate(zip(eng.wip_forward_h(sc, ps), ps.items)):
            ws, wskl, wflag = oracle.wip_forward_h(sc, p)
            assert s == ws and flag == flagmap[wflag], (seed, rnd, i)
            if wflag == 0:
                assert skl.tolist() == wskl.tolist(), (seed, rnd, i)


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_protein_udh(eng, seed):
    from oracle import oracle
    rng = np.random.default_rng(synth.SEED + 9200 + seed)
    n_empty = n_full = 0
    for rnd in range(4):
        sc = _rand_scoring_h(rng)
        ps = abi.ProblemSetH()
        while len(ps) < 24:
            p = _rand_problem_h(rng, ps)
            if p.a_right - p.a_left < 34:
                ps.items.pop()
        n_im = int(rng.integers(1, 3))
        us, ucpos, urng = eng.wip_udh_h(sc, ps, n_im)
        for i, p in enumerate(ps.items):
            ws, wcpos, wrng = oracle.wip_udh_h(sc, p, n_im)
            fs, fskl, fflag = oracle.wip_forward_h(sc, p)
            if fflag != 0 or not _well_defined(wrng, wcpos, fskl, 3):
                n_empty += 1
                continue
            n_full += 1
            assert int(us[i]) == ws and urng[i].tolist() == wrng.tolist(), (seed, rnd, i)
            assert ucpos[i].tolist() == wcpos.tolist(), (seed, rnd, i)
    assert n_full > n_empty


def _ladder_udh_n_im(sc, m, n, step):
    """n_imd lspS_ng / lspH_ng would pick (0: traceback branch, -1: recursive)"""
    import math
    coef_b, coef_c = 2.0, 12.0
    cvol = float(m) * (n + step * m)
    if abs(n - m) < (8 if step == 1 else 16) or coef_b * cvol < sc.max_vmf_space:
        return 0
    imd1 = int(math.pow(2.0 * m * coef_b / coef_c, 1.0 / 3) + 0.5) - 1
    if coef_c * n * imd1 + coef_b * cvol / (imd1 + 1) / (imd1 + 1) > sc.max_vmf_space:
        return -1
    n_imd = sc.ubh if sc.ubh else min(imd1, m // 16)
    if ((m + n_imd) // (n_imd + 1)) * n_imd == m:
        n_imd -= 1
    return n_imd


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_cdna_ladder(eng, seed):
    """alignS_ng with a small MaxVmfSpace (linear-space branch, slabs, recursion) under random parameters: GPU ladder ==
    oracle ladder for EVERY query the library does not mark itself (SpdpAlignment.flags & SPDP_ALN_LEFT_EDGE: the path of
    a linear-space call ran along the free left edge, where the reference reads link lanes it never initialised);
    nothing is skipped on the oracle's say-so"""
    from oracle import oracle, host_logic
    rng = np.random.default_rng(synth.SEED + 9300 + seed)
    n_cmp = n_marked = 0
    for rnd in range(3):
        sc = _rand_scoring_s(rng)
        sc.max_vmf_space = int(rng.choice([3000, 8000, 20000, 60000]))
        ps = abi.ProblemSet()
        for _ in range(40):
            _rand_problem_s(rng, ps)
        res = eng.align_s(sc, ps, allow_partial=True, with_flags=True)
        for p, (score, skl, flags) in zip(ps.items, res):
            m, n = p.a_right - p.a_left, p.b_right - p.b_left
            k = _ladder_udh_n_im(sc, m, n, 1)
            marked = bool(flags & abi.ALN_LEFT_EDGE)
            if marked:
                n_marked += 1
                continue
            try:
                wscr, wskl = host_logic.align_s(sc, p)
            except host_logic.NeedsScalarEngine:
                continue
            n_cmp += 1
            assert score == wscr and skl.ravel().tolist() == (wskl or [])
    assert n_cmp > 2 * n_marked and n_cmp >= 60


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_protein_ladder(eng, seed):
    """alignH_ng with a small MaxVmfSpace under random parameters, as test_fuzz_cdna_ladder"""
    from oracle import oracle, host_logic_h as hh
    rng = np.random.default_rng(synth.SEED + 9400 + seed)
    n_cmp = n_skip = 0
    for rnd in range(3):
        sc = _rand_scoring_h(rng)
        sc.max_vmf_space = int(rng.choice([20000, 60000, 200000]))
        ps = abi.ProblemSetH()
        for _ in range(32):
            _rand_problem_h(rng, ps)
        res = eng.align_h(sc, ps)
        for p, (score, skl, flag) in zip(ps.items, res):
            m, n = p.a_right - p.a_left, p.b_right - p.b_left
            k = _ladder_udh_n_im(sc, m, n, 3)
            if k != 0 and m >= 17:
                ws, wcpos, wrng = oracle.wip_udh_h(sc, p, max(k, 1))
                fs, fskl, fflag = oracle.wip_forward_h(sc, p)
                if fflag != 0 or not _well_defined(wrng, wcpos, fskl, 3):
                    n_skip += 1
                    continue
            try:
                wscr, wskl = hh.align_h(sc, p)
                wflag = 0
            except hh.ReferenceUndefined:
                wflag = -2
            except hh.ReferenceFatal:
                wflag = -1
            except hh.NotRestated:
                wflag = 1
            n_cmp += 1
            assert flag == wflag
            if wflag == 0:
                assert score == wscr and skl.ravel().tolist() == (wskl or [])
    assert n_cmp > n_skip


# ---- the -A0 wavefront kernels (spdp_rowwave.hip / spdp_h_rowwave.hip) under fully random inputs -------------------
def _rand_exact_s(rng):
    """random length-penalty table, junction table and per-position classes on top of _rand_scoring_s"""
    sc0 = _rand_scoring_s(rng)
    intpen = (-rng.integers(100, 500, size=1200)).astype(np.int16)
    intpen[:int(rng.integers(5, 60))] = -32768 + 1024
    t53 = rng.integers(-80, 40, size=256).astype(np.int16)
    kw = dict(gop=sc0.gop, gep=sc0.gep, ipen=sc0.ipen, llmt=sc0.llmt, qm_len=list(sc0.qm_len)[:5], qm_pen=list(sc0.qm_pen)[:5],
              nquant=sc0.nquant, sh=sc0.sh, intpen=intpen, t53=t53, scalar_engines=1)
    return defaults.scoring(**kw)


def _with_classes(rng, ps):
    out = abi.ProblemSet()
    for p in ps.items:
        n = p.b_len + 1
        c5 = (rng.random(n) < 0.08).astype(np.uint8); c3 = (rng.random(n) < 0.08).astype(np.uint8)
        dinc = rng.integers(0, 256, size=n).astype(np.uint8)
        a = np.ctypeslib.as_array(C_cast_u8(p.a), shape=(p.a_len,)).copy()
        b = np.ctypeslib.as_array(C_cast_u8(p.b), shape=(p.b_len,)).copy()
        s5 = np.ctypeslib.as_array(C_cast_i16(p.sig5), shape=(n,)).copy()
        s3 = np.ctypeslib.as_array(C_cast_i16(p.sig3), shape=(n,)).copy()
        out.add(a, b, s5, s3, p.a_left, p.a_right, p.b_left, p.b_right, (p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr),
                cano5=c5, cano3=c3, dinc=dinc)
    return out


def C_cast_u8(ptr):
    import ctypes as C
    return C.cast(ptr, C.POINTER(C.c_uint8))


def C_cast_i16(ptr):
    import ctypes as C
    return C.cast(ptr, C.POINTER(C.c_int16))


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_cdna_a0_engines(eng, seed):
    """scorealoneS_ng, forwardS_ng (records) and hirschbergS_ng (cpos rows) as wavefront kernels against the oracle:
    random gap / intron parameters, signals, donor / acceptor sites, classes, length penalties, sub-ranges, end flags"""
    from oracle import oracle
    rng = np.random.default_rng(synth.SEED + 9500 + seed)
    n_udh = 0
    for rnd in range(3):
        sc = _rand_exact_s(rng)
        base = abi.ProblemSet()
        for _ in range(32):
            _rand_problem_s(rng, base)
        ps = _with_classes(rng, base)
        assert eng.scalar_scorealone(sc, ps).tolist() == [oracle.scalar_scorealone(sc, p) for p in ps.items]
        for (s, skl), p in zip(eng.scalar_forward(sc, ps), ps.items):
            ws, wskl = oracle.scalar_forward(sc, p)
            assert s == ws and skl.tolist() == wskl.tolist()
        for n_im in (1, 2):
            big = abi.ProblemSet()
            big._keep = ps._keep
            m = 40 + 6 * n_im
            for p in ps.items:
                if p.a_right - p.a_left >= m:
                    q = abi.Problem.from_buffer_copy(p)
                    q.a_right = q.a_left + m                   # one imd_intvl for the batch
                    big.items.append(q)
            if not len(big):
                continue
            intvl = (m + n_im) // (n_im + 1)
            scores, cpos, ranges, flags = eng.scalar_udh(sc, big, n_im, intvl)
            for i, p in enumerate(big.items):
                ws, wcpos, wrng, wflag = oracle.scalar_udh(sc, p, n_im, intvl)
                assert int(flags[i]) == wflag, (seed, rnd, n_im, i)
                if wflag == 0:
                    assert int(scores[i]) == ws and ranges[i].tolist() == wrng.tolist() and cpos[i].tolist() == wcpos.tolist(), \
                        (seed, rnd, n_im, i)
                    n_udh += 1
    assert n_udh >= 30


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_protein_a0_engines(eng, seed):
    """forwardH_ng (score-only and records) and hirschbergH_ng as wavefront kernels against the oracle under random
    parameters, signals, phases, tables and sub-ranges"""
    from oracle import oracle
    rng = np.random.default_rng(synth.SEED + 9600 + seed)
    n_udh = 0
    for rnd in range(3):
        sc0 = _rand_scoring_h(rng)
        intpen = (-rng.integers(100, 500, size=1500)).astype(np.int16)
        t53 = rng.integers(-80, 40, size=256).astype(np.int16)
        sc = defaults.scoring_h(gop=sc0.gop, gep=sc0.gep, gapw1=sc0.gapw1, gapw2=sc0.gapw2, gapw3=sc0.gapw3, ipen=sc0.ipen,
                                llmt=sc0.llmt, qm_len=list(sc0.qm_len)[:5], qm_pen=list(sc0.qm_pen)[:5], nquant=sc0.nquant,
                                sh=sc0.sh, term_codon=sc0.term_codon, intpen=intpen, t53=t53, scalar_engines=1,
                                minl=int(rng.integers(20, 60)), gape1=-int(rng.integers(100, 400)),
                                gape2=-int(rng.integers(100, 400)), extragop=-int(rng.integers(0, 200)))
        ps = abi.ProblemSetH()
        for _ in range(24):
            p = _rand_problem_h(rng, ps)
            dc = rng.integers(0, 256, size=p.b_len + 3).astype(np.uint8)
            ps._keep.append(dc)
            p.dinc = dc.ctypes.data
            ps.items[-1] = p
        for tb in (False, True):
            res = eng.scalar_forward_h(sc, ps, traceback=tb)
            for (s, skl), p in zip(res, ps.items):
                ws, wskl = oracle.scalar_forward_h(sc, p, traceback=tb)
                assert s == ws, (seed, rnd, tb)
                if tb:
                    assert skl.tolist() == wskl.tolist(), (seed, rnd)
        n_im, m = 1, 36
        big = abi.ProblemSetH()
        big._keep = ps._keep
        for p in ps.items:
            if p.a_right - p.a_left >= m:
                q = abi.ProblemH.from_buffer_copy(p)
                q.a_right = q.a_left + m
                big.items.append(q)
        if len(big):
            intvl = (m + n_im) // (n_im + 1)
            scores, cpos, ranges, flags = eng.scalar_udh_h(sc, big, n_im, intvl)
            for i, p in enumerate(big.items):
                ws, wcpos, wrng, wflag = oracle.scalar_udh_h(sc, p, n_im, intvl)
                assert int(flags[i]) == wflag, (seed, rnd, i)
                if wflag == 0:
                    assert int(scores[i]) == ws and ranges[i].tolist() == wrng.tolist() and cpos[i].tolist() == wcpos.tolist(), \
                        (seed, rnd, i)
                    n_udh += 1
    assert n_udh >= 12
