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
