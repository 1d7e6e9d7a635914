"""The two generations of the `_wip` sweeps must agree bit for bit: spdp_sweep_fp (fp32-issue form, the
default for non-local score-only / linear-space runs) against spdp_sweep (int32, SPDP_FP=0), on the
engine entry points and through the whole alignS_ng ladder; and a batch run as pipelined chunks on
lanes of the context (SPDP_CHUNKS) against the same batch in one piece.  Both knobs are read per call."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update({k: str(v) for k, v in self.kv.items()})

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _batch(n, seed, **kw):
    from spaln_amd import abi, synth
    ps = abi.ProblemSet()
    for w, q, s5, s3, _ in synth.make_batch(n, seed=seed, **kw):
        ps.add(q, w, s5, s3)
    return ps


@pytest.mark.parametrize("shape", [dict(mrna_len=700, n_exons=5, flank=400, intron_hi=1500),
                                   dict(mrna_len=2000, n_exons=8, flank=1000),
                                   dict(mrna_len=333, n_exons=3, flank=77, intron_hi=900)])
def test_fp_sweep_equals_int_sweep_engines(shape):
    from spaln_amd import defaults, engine
    sc = defaults.scoring()
    ps = _batch(40, 4242, **shape)
    eng = engine.Engine(0)
    out = {}
    for fp in (0, 1):
        with _Env(SPDP_FP=fp):
            s = eng.wip_scoreonly(sc, ps)
            us, ucpos, urng = eng.wip_udh(sc, ps, 5)
            fw = [(int(r[0]), r[1].tolist()) for r in eng.wip_forward(sc, ps)]      # (the traceback flavour: round 4)
            out[fp] = (s.tolist(), us.tolist(), ucpos.tolist(), urng.tolist(), fw)
    eng.close()
    assert out[0] == out[1]


@pytest.mark.parametrize("variant", ["default", "flat", "noll_spj_off", "free_ends_off"])
def test_fp_sweep_equals_int_sweep_variants(variant):
    """parameter sets that take other paths through the kernel: the flat -A3 penalty (nquant = 1), splice
    signals off, global ends"""
    from spaln_amd import defaults, engine
    sc = defaults.scoring()
    ps = _batch(24, 99, mrna_len=900, n_exons=4, flank=300, intron_hi=2000)
    if variant == "flat":
        sc.nquant = 1
    if variant == "noll_spj_off":
        sc.spj = 0
    if variant == "free_ends_off":
        for p in ps.items:
            p.a_exgl = p.a_exgr = p.b_exgl = p.b_exgr = 0
    eng = engine.Engine(0)
    out = {}
    for fp in (0, 1):
        with _Env(SPDP_FP=fp):
            s = eng.wip_scoreonly(sc, ps)
            us, ucpos, urng = eng.wip_udh(sc, ps, 3)
            fw = [(int(r[0]), r[1].tolist()) for r in eng.wip_forward(sc, ps)]
            out[fp] = (s.tolist(), us.tolist(), ucpos.tolist(), urng.tolist(), fw)
    eng.close()
    assert out[0] == out[1]


def test_fp_sweep_ladder_and_chunks():
    """alignS_ng over 400 C2-sized queries: int sweeps in one piece = fp sweeps in one piece = fp sweeps as 3 chunks"""
    from spaln_amd import defaults, engine
    sc = defaults.scoring()
    ps = _batch(400, 31337)
    eng = engine.Engine(0)
    res = {}
    for name, env in (("int", dict(SPDP_FP=0, SPDP_CHUNKS=1)), ("fp", dict(SPDP_FP=1, SPDP_CHUNKS=1)),
                      ("fp3", dict(SPDP_FP=1, SPDP_CHUNKS=3))):
        with _Env(**env):
            res[name] = [(s, skl.tolist()) for s, skl in eng.align_s(sc, ps)]
    eng.close()
    assert res["int"] == res["fp"]
    assert res["fp"] == res["fp3"]
    with _Env(SPDP_FP=1, SPDP_FP_FWD=0, SPDP_CHUNKS=1):                           # fp sweeps, the traceback sweep on spdp_kernels.hip
        eng2 = engine.Engine(0)
        mixed = [(s, skl.tolist()) for s, skl in eng2.align_s(sc, ps)]
        eng2.close()
    assert mixed == res["fp"]
    assert sum(1 for s, skl in res["fp"] if len(skl) > 3) > 360


def test_fp_sweep_long_query_cross_cu():
    """one long cDNA (the top of the recursion runs as cross-CU pass pipelines): fp = int"""
    from spaln_amd import abi, defaults, engine, synth
    sc = defaults.scoring()
    ps = abi.ProblemSet()
    for w, q, s5, s3, _ in synth.make_batch(1, seed=5, mrna_len=9000, n_exons=12, flank=1000, intron_hi=3000):
        ps.add(q, w, s5, s3)
    eng = engine.Engine(0)
    res = {}
    for fp in (0, 1):
        with _Env(SPDP_FP=fp):
            res[fp] = [(s, skl.tolist()) for s, skl in eng.align_s(sc, ps)]
            us, ucpos, urng = eng.wip_udh(sc, ps, 7)
            res[fp].append((us.tolist(), ucpos.tolist(), urng.tolist()))
    eng.close()
    assert res[0] == res[1]


def test_tall_slabs_on_the_side_stream_and_across_cus():
    """two 50 kb cDNAs (BASELINE config 5): the recursion leaves traceback slabs of thousands of rows; they run as a launch of
    their own beside the short ones, each as a cross-CU pass pipeline (spdp_sweep_fp<FL_FORWARD, ., CROSS>).  Same records as
    one launch (SPDP_SPLIT_FWD=0), as 16-wave blocks (SPDP_CROSS=0), and as the int32 sweeps (SPDP_FP=0)"""
    from spaln_amd import abi, defaults, engine, synth
    sc = defaults.scoring()
    ps = abi.ProblemSet()
    # (the bench's own batch: two of its 32 queries leave a slab of 17 528 rows in a band of 13 361 diagonals)
    for w, q, s5, s3, _ in synth.make_batch(32, seed=synth.SEED + 55, n_exons=25, mrna_len=50000, flank=1000, intron_lo=1000, intron_hi=10000):
        ps.add(q, w, s5, s3)
    res = {}
    for name, env in (("default", {}), ("one_launch", dict(SPDP_SPLIT_FWD=0)), ("no_cross", dict(SPDP_CROSS=0)), ("int", dict(SPDP_FP=0)),
                      ("gave_up", dict(SPDP_CROSS_TEST_SHORT=1))):       # a block of every cross-CU launch never arrives: the launches are repeated
        with _Env(**env):
            eng = engine.Engine(0)
            res[name] = [(s, skl.tolist()) for s, skl in eng.align_s(sc, ps)]
            eng.close()
    assert sum(1 for _, skl in res["default"] if len(skl) > 40) >= 30
    for name in ("one_launch", "no_cross", "int", "gave_up"):
        assert res[name] == res["default"], name
