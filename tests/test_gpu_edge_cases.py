"""Edge cases of the C ABI on the GPU: empty batches, ragged batches, problems that need an engine
that is not built, invalid inputs (loud errors, never a silent fallback)."""
import ctypes as C

import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi, defaults, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


def test_empty_batches(eng):
    sc = defaults.scoring()
    ps = abi.ProblemSet()
    assert eng.wip_scoreonly(sc, ps).size == 0
    assert eng.homscore_s(sc, ps).size == 0
    assert eng.wip_forward(sc, ps) == []
    assert eng.align_s(sc, ps) == []
    sch = defaults.scoring_h()
    psh = abi.ProblemSetH()
    assert eng.wip_forward_h(sch, psh) == []
    assert eng.align_h(sch, psh) == []
    assert eng.homscore_h(sch, psh).size == 0


def test_ragged_protein_batch(eng):
    """one launch holding 8-residue to 400-residue queries, every band shape"""
    from oracle import oracle
    files = [f for f in golden_files("h1_") if "local" not in f
             and not any(f.endswith(f"tiny_m{m}.spdg") for m in (3, 5, 7))]     # forwardH1_wip needs >= 8 rows
    sc = spdg.scoring_h(spdg.load(files[0]))
    ps = abi.ProblemSetH()
    for f in files:
        spdg.problem_h(spdg.load(f), ps)
    res = eng.wip_forward_h(sc, ps)
    for p, (score, skl, flag) in zip(ps.items, res):
        s, oskl, oflag = oracle.wip_forward_h(sc, p)
        assert score == s and flag == {0: 0, -2: -1, -3: -2}[oflag]
        if oflag == 0:
            assert skl.tolist() == oskl.tolist()


def test_short_queries_are_flagged_not_faked(eng):
    """fewer than 8 residues: the reference switches to its scalar engine; without that engine's inputs
    (intpen / t53 are there, dinc is not) the problem comes back flagged, the rest of the batch is
    computed; with them it is computed (tests/test_gpu_scalar_h.py)"""
    fx = spdg.load([f for f in golden_files("h1_") if f.endswith("h1_basic.spdg")][0])
    sc = spdg.scoring_h(fx)
    q = fx["prm"]
    ps = abi.ProblemSetH()
    for ar in (5, 7, q["a_right"]):
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
               fx["phs5"], fx["phs3"], 0, ar, q["b_left"], q["b_right"], (1, 1, 1, 1))
    res = eng.align_h(sc, ps)
    assert [r[2] for r in res] == [1, 1, 0]
    assert res[0][0] == abi.NEVSEL and res[0][1].size == 0
    assert res[2][1].ravel().tolist() == fx["aln_skl_A2"].tolist()
    # the same batch with dinc: nothing is flagged, the short ones equal the oracle's scalar ladder
    from oracle import host_logic_h as hh
    dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
    ps = abi.ProblemSetH()
    for ar in (5, 7, q["a_right"]):
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
               fx["phs5"], fx["phs3"], 0, ar, q["b_left"], q["b_right"], (1, 1, 1, 1), dinc=dinc)
    res = eng.align_h(sc, ps)
    assert [r[2] for r in res] == [0, 0, 0]
    for p, (score, skl, _) in zip(ps.items, res):
        ws, wskl = hh.align_h(sc, p)
        assert score == ws and skl.ravel().tolist() == (wskl or [])


def test_invalid_inputs_raise(eng):
    fx = spdg.load([f for f in golden_files("h1_") if f.endswith("h1_basic.spdg")][0])
    sc = spdg.scoring_h(fx)
    q = fx["prm"]

    def one(**kw):
        ps = abi.ProblemSetH()
        args = dict(a_left=0, a_right=q["a_right"], b_left=0, b_right=q["b_right"])
        args.update(kw)
        exin = args.pop("exin", None)
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], fx["sigS"], fx["sigT"], fx["sigE"],
               fx["phs5"], fx["phs3"], args["a_left"], args["a_right"], args["b_left"], args["b_right"],
               (1, 1, 1, 1), exin=exin)
        return ps

    with pytest.raises(RuntimeError, match="query range"):
        eng.wip_forward_h(sc, one(a_right=q["a_right"] + 5))
    with pytest.raises(RuntimeError, match="genomic range"):
        eng.wip_forward_h(sc, one(b_right=q["b_right"] + 50))
    with pytest.raises(RuntimeError, match="Exinon"):
        eng.wip_forward_h(sc, one(exin=(100, q["b_right"])))
    bad = spdg.scoring_h(fx, nquant=1)
    bad.nquant = 0
    with pytest.raises(RuntimeError, match="nquant"):
        eng.wip_forward_h(bad, one())


def test_mixed_cdna_batch_sizes(eng):
    """30-nt to 1450-nt cDNAs in one align call equal the per-problem results"""
    files = [f for f in golden_files("s1_") if "local" not in f and "tiny_m1" not in f and "tiny_m3" not in f
             and "tiny_m7" not in f]
    sc = spdg.scoring(spdg.load(files[0]))
    together = abi.ProblemSet()
    for f in files:
        spdg.problem(spdg.load(f), together)
    res_all = eng.wip_scoreonly(sc, together)
    for i, f in enumerate(files):
        ps, _ = spdg.problem(spdg.load(f))
        assert int(eng.wip_scoreonly(sc, ps)[0]) == int(res_all[i])


def test_submit_wait_two_contexts(eng):
    """spdp_submit_align_s / spdp_wait: two contexts on one GPU with batches in flight at the same time,
    same results as the synchronous call"""
    from spaln_amd import engine, synth
    sc = defaults.scoring()
    sets = []
    for seed in (3, 4):
        ps = abi.ProblemSet()
        for w, q, s5, s3, _ in synth.make_batch(24, seed=seed, n_exons=4, mrna_len=600, flank=300, intron_hi=1500):
            ps.add(q, w, s5, s3)
        sets.append(ps)
    want = [eng.align_s(sc, ps) for ps in sets]
    e2 = engine.Engine(0)
    try:
        waits = [eng.submit_align_s(sc, sets[0]), e2.submit_align_s(sc, sets[1])]
        got = [w() for w in waits]
        assert all(w.poll() for w in waits)
    finally:
        e2.close()
    for g, w in zip(got, want):
        assert [(s, k.tolist()) for s, k in g] == [(s, k.tolist()) for s, k in w]


def test_skl_slot_overflow_is_walked_again(monkeypatch):
    """traceback record lists longer than their device slot (SPDP_SKL_CAP shrinks the slots to 6 records): the walk is
    repeated for those problems with full-size slots and the batch comes back complete and unchanged"""
    from spaln_amd import abi, defaults, engine, synth
    sc = defaults.scoring()
    ps = abi.ProblemSet()
    for w, q, s5, s3, _ in synth.make_batch(12, seed=77, mrna_len=600, n_exons=5, flank=200, intron_hi=900, indel=0.02):
        ps.add(q, w, s5, s3)
    eng = engine.Engine(0)
    want = [(s, skl.tolist()) for s, skl in eng.align_s(sc, ps)]
    want_f = [(s, skl.tolist()) for s, skl in eng.wip_forward(sc, ps)]
    assert max(len(skl) for _, skl in want_f) > 8
    monkeypatch.setenv("SPDP_SKL_CAP", "6")
    got = [(s, skl.tolist()) for s, skl in eng.align_s(sc, ps)]
    got_f = [(s, skl.tolist()) for s, skl in eng.wip_forward(sc, ps)]
    eng.close()
    assert got == want and got_f == want_f
