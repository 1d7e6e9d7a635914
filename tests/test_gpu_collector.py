"""SpdpCollector -- one-problem calls from many host threads (the reference's seeded walks under its
thread pool) run as device batches: results equal the direct batch call, and batches really form."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problems(n, seed):
    from spaln_amd import abi, synth
    ps = abi.ProblemSet()
    for w, q, s5, s3, _ in synth.make_batch(n, seed=seed, mrna_len=600, n_exons=4, flank=200, intron_hi=900):
        ps.add(q, w, s5, s3)
    return ps


def test_lsp_records_give_the_alignment():
    """spdp_lsp_s hands over the records lspS_ng writes; stdskl + trimskl of them is spdp_align_s's corner list"""
    from spaln_amd import defaults, engine
    from oracle import host_logic
    sc = defaults.scoring()
    ps = _problems(24, 5)
    eng = engine.Engine(0)
    raw = eng.lsp_s(sc, ps)
    full = eng.align_s(sc, ps)
    eng.close()
    for p, (s_raw, rec), (s_full, skl) in zip(ps.items, raw, full):
        assert s_raw == s_full
        std = host_logic.trim_skl(host_logic.std_skl([tuple(map(int, r)) for r in rec]), p)
        assert [list(map(int, r)) for r in std] == skl[1:].tolist()


@pytest.mark.parametrize("raw", [False, True])
def test_collector_equals_direct_batch(raw):
    from spaln_amd import defaults, engine
    sc = defaults.scoring()
    n, n_threads = 192, 48
    ps = _problems(n, 11)
    eng = engine.Engine(0)
    want = eng.lsp_s(sc, ps) if raw else eng.align_s(sc, ps)
    col = engine.Collector(eng, sc, max_batch=64, max_wait_us=20000, raw=raw)
    got = [None] * n
    errs = []

    def worker(t):
        try:
            for i in range(t, n, n_threads):           # each thread walks "its" queries one call at a time
                got[i] = col.align_s(ps.items[i])
        except Exception as e:                          # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    st = col.stats()
    col.close()
    eng.close()
    assert not errs, errs[:2]
    for i in range(n):
        assert got[i][0] == want[i][0], i
        a, b = got[i][1].tolist(), want[i][1].tolist()
        assert (sorted(a) == sorted(b)) if raw else (a == b), i
    assert st["requests"] == n
    assert st["batches"] < n // 4 and st["largest"] >= 16, st     # 48 threads in flight: batches of tens, not of one


def test_collector_single_caller_times_out_into_a_batch_of_one():
    from spaln_amd import defaults, engine
    sc = defaults.scoring()
    ps = _problems(3, 3)
    eng = engine.Engine(0)
    want = eng.align_s(sc, ps)
    col = engine.Collector(eng, sc, max_batch=64, max_wait_us=500)
    got = [col.align_s(p) for p in ps.items]
    st = col.stats()
    col.close()
    eng.close()
    assert [(s, k.tolist()) for s, k in got] == [(s, k.tolist()) for s, k in want]
    assert st["batches"] == 3 and st["largest"] == 1


def _problems_h(n, seed):
    from spaln_amd import abi, synth
    ps = abi.ProblemSetH()
    for g, sg in synth.make_protein_batch(n, seed=seed, aa_len=150, n_exons=3, flank=200, intron_hi=600):
        ps.add(synth.encode_protein(g.query), sg["b"], sg["sig5"], sg["sig3"], sg["sigS"], sg["sigT"], sg["sigE"],
               sg["phs5"], sg["phs3"])
    return ps


def test_lsp_h_records_give_the_alignment():
    """spdp_lsp_h hands over the records lspH_ng writes (the protein side of the seeded path's DP call); stdskl3 of them
    is spdp_align_h's corner list, and the oracle's ladder writes the same records"""
    from spaln_amd import defaults, engine
    from oracle import host_logic_h as hh, oracle
    sc = defaults.scoring_h(max_vmf_space=400000)           # part of the batch through the linear-space branch
    ps = _problems_h(16, 5)
    eng = engine.Engine(0)
    raw = eng.lsp_h(sc, ps)
    full = eng.align_h(sc, ps)
    eng.close()
    n_ok = 0
    for p, (s_raw, rec, f_raw), (s_full, skl, f_full) in zip(ps.items, raw, full):
        assert (s_raw, f_raw) == (s_full, f_full)
        if f_raw:
            continue
        std = hh.std_skl3([tuple(map(int, r)) for r in rec]) if len(rec) >= 2 else []
        assert [list(map(int, r)) for r in std] == skl[1:].tolist()
        wrec = []
        try:
            ws = hh.lsp_h(sc, p, oracle.stripe31(p, sc.sh), wrec)
        except (hh.ReferenceUndefined, hh.ReferenceFatal):
            continue
        assert ws == s_raw and sorted(map(tuple, rec.tolist())) == sorted(wrec)
        n_ok += 1
    assert n_ok >= 12


@pytest.mark.parametrize("raw", [False, True])
def test_protein_collector_equals_direct_batch(raw):
    from spaln_amd import defaults, engine
    sc = defaults.scoring_h()
    n, n_threads = 96, 24
    ps = _problems_h(n, 11)
    eng = engine.Engine(0)
    want = eng.lsp_h(sc, ps) if raw else eng.align_h(sc, ps)
    col = engine.Collector(eng, sc, max_batch=48, max_wait_us=20000, raw=raw)
    got = [None] * n
    errs = []

    def worker(t):
        try:
            for i in range(t, n, n_threads):
                got[i] = col.align_h(ps.items[i])
        except Exception as e:                          # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    st = col.stats()
    col.close()
    eng.close()
    assert not errs, errs
    for (ws, wk, wf), (gs, gk, gf) in zip(want, got):
        assert (ws, wf) == (gs, gf) and wk.tolist() == gk.tolist()
    assert st["requests"] == n and st["batches"] < n and st["largest"] > 1
