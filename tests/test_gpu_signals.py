"""SURVEY 8 f1 on the device: spdp_splice_signals (spdp_signals.hip) against the reference's arrays in the
fixtures and against the oracle; and a batch uploaded as plain codes (SpdpScoring::sigmodel, sig5 = NULL) against
the same batch with the arrays supplied by the host -- `_wip` ladder and the exact engines."""
import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files, golden_ids
from tests.test_oracle_signals import FILES, IDS, _stale

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", FILES, ids=IDS)
def test_device_signals_equal_reference(path):
    from spaln_amd import abi, engine
    fx = spdg.load(path)
    model = abi.signal_model_from_fixture(fx)
    b = fx["b_codes"]
    left, right = fx["prm"]["b_left"], fx["prm"]["b_right"]
    ok5, ok3 = _stale(b.size, left, right)
    eng = engine.Engine(0)
    got = eng.splice_signals(model, b, left, right)
    eng.close()
    assert np.array_equal(got["sig5"][ok5], fx["sig5"][ok5])
    assert np.array_equal(got["sig3"][ok3], fx["sig3"][ok3])
    assert np.array_equal(got["dinc"] >> 4, fx["dinc5"]) and np.array_equal(got["dinc"] & 15, fx["dinc3"])
    assert np.array_equal((got["cano5"] > 0)[ok5], (fx["cano5"] > 0)[ok5])
    assert np.array_equal((got["cano3"] > 0)[ok3], (fx["cano3"] > 0)[ok3])


def test_device_signals_equal_oracle_everywhere():
    """including the cells the reference leaves stale, windows with Ns, sub-ranges, tiny windows"""
    from oracle import signals
    from spaln_amd import abi, engine
    fx = spdg.load(golden_files("s1_basic")[0])
    md = signals.model_of(fx)
    model = abi.signal_model_from_fixture(fx)
    rng = np.random.default_rng(7)
    eng = engine.Engine(0)
    cases = []
    b = fx["b_codes"][:1500].copy()
    cases.append((b, 0, b.size))
    b2 = b.copy(); b2[rng.integers(0, b2.size, 12)] = 15          # N
    cases += [(b2, 0, b2.size), (b2, 300, 1100), (b[:5], 0, 5), (b[:30], 2, 29), (b[:1], 0, 1), (b[:300], 256, 300)]
    for bb, lo, hi in cases:
        got = eng.splice_signals(model, bb, lo, hi)
        s5, s3 = signals.splice_signals(md, bb, lo, hi)
        d5, d3, c5, c3 = signals.classes(bb, lo, hi, md["any"], md["both_ori"])
        assert np.array_equal(got["sig5"], s5) and np.array_equal(got["sig3"], s3), (bb.size, lo, hi)
        assert np.array_equal(got["dinc"], (d5 << 4) | d3)
        assert np.array_equal(got["cano5"], c5) and np.array_equal(got["cano3"], c3)
    eng.close()


def _sets(n, seed, model_src, eng, **kw):
    """the same synthetic batch twice: arrays supplied (computed here with spdp_splice_signals), and codes only"""
    from spaln_amd import abi, synth
    with_arrays, codes_only = abi.ProblemSet(), abi.ProblemSet()
    for w, q, _s5, _s3, _ in synth.make_batch(n, seed=seed, **kw):
        sg = eng.splice_signals(model_src, w)
        with_arrays.add(q, w, sg["sig5"], sg["sig3"], cano5=sg["cano5"], cano3=sg["cano3"], dinc=sg["dinc"])
        codes_only.add(q, w, None, None)
    return with_arrays, codes_only


@pytest.mark.parametrize("engines", [0, 1, 2])
def test_batch_from_codes_equals_batch_from_arrays(engines):
    from spaln_amd import abi, engine
    fx = spdg.load(golden_files("s1_basic")[0])
    model = abi.signal_model_from_fixture(fx)
    eng = engine.Engine(0)
    n = 48 if engines == 0 else 12
    shape = dict(mrna_len=900, n_exons=5, flank=300, intron_hi=1500) if engines else {}
    pa, pc = _sets(n, 2024 + engines, model, eng, **shape)
    sc_a = spdg.scoring(fx, scalar_engines=engines)
    sc_c = spdg.scoring(fx, scalar_engines=engines, sigmodel=model)
    ra = [(s, skl.tolist()) for s, skl in eng.align_s(sc_a, pa)]
    rc = [(s, skl.tolist()) for s, skl in eng.align_s(sc_c, pc)]
    ha = eng.homscore_s(sc_a, pa).tolist()
    hc = eng.homscore_s(sc_c, pc).tolist()
    eng.close()
    assert ra == rc and ha == hc
    assert sum(1 for s, skl in rc if len(skl) > 3) >= n - 2          # real spliced alignments came out


def test_codes_only_without_model_is_refused():
    from spaln_amd import abi, engine, synth
    fx = spdg.load(golden_files("s1_basic")[0])
    ps = abi.ProblemSet()
    for w, q, _s5, _s3, _ in synth.make_batch(2, seed=1, mrna_len=300, n_exons=2, flank=100, intron_hi=400):
        ps.add(q, w, None, None)
    eng = engine.Engine(0)
    with pytest.raises(RuntimeError, match="sigmodel"):
        eng.homscore_s(spdg.scoring(fx), ps)
    eng.close()


def test_fixture_alignment_from_codes():
    """reference fixtures whose arrays hold nothing stale: the reference's own alignment from codes + model alone"""
    from spaln_amd import abi, engine
    eng = engine.Engine(0)
    n = 0
    for path in golden_files("s1_"):
        fx = spdg.load(path)
        q = fx["prm"]
        if "aln_scr_A2" not in fx or q["local"]:
            continue
        model = abi.signal_model_from_fixture(fx)
        got = eng.splice_signals(model, fx["b_codes"], q["b_left"], q["b_right"])
        if not (np.array_equal(got["sig5"], fx["sig5"]) and np.array_equal(got["sig3"], fx["sig3"])):
            continue                                    # a stale boundary cell in the reference's run (see the oracle test)
        ps = abi.ProblemSet()
        ps.add(fx["a_codes"], fx["b_codes"], None, None, q["a_left"], q["a_right"], q["b_left"], q["b_right"],
               (q["a_exgl"], q["a_exgr"], q["b_exgl"], q["b_exgr"]))
        res = eng.align_s(spdg.scoring(fx, sigmodel=model), ps)
        want = int(fx["aln_scr_A2"][0])
        assert res[0][0] == want, path
        assert res[0][1].ravel().tolist() == fx["aln_skl_A2"].tolist(), path
        n += 1
    eng.close()
    assert n >= 12


@pytest.mark.parametrize("engines", [0, 1])
def test_subrange_problem_from_codes_reads_the_gene_windows_signals(engines):
    """an engine call on a SUB-range (what lspS_ng gets between two HSPs) reads the signals of the Exinon built over the
    whole gene window: with SpdpProblem.exin_left / exin_right a codes-only upload equals the upload of the arrays, and
    without them (signals recomputed from the sub-range's own left end) it need not"""
    from spaln_amd import abi, engine
    eng = engine.Engine(0)
    n_diff = 0
    for name in ("s1_basic", "s1_1400nt", "s1_indels", "s1_divergent"):
        fx = spdg.load([f for f in golden_files("s1_") if f.endswith(name + ".spdg")][0])
        q = fx["prm"]
        model = abi.signal_model_from_fixture(fx)
        got = eng.splice_signals(model, fx["b_codes"], q["b_left"], q["b_right"])
        if not (np.array_equal(got["sig5"], fx["sig5"]) and np.array_equal(got["sig3"], fx["sig3"])):
            continue
        m, n = q["a_right"] - q["a_left"], q["b_right"] - q["b_left"]
        sub = (q["a_left"] + m // 5, q["a_right"] - m // 4, q["b_left"] + n // 6 + 1, q["b_right"] - n // 7)
        dinc = (fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8")
        arrays, codes, naive = abi.ProblemSet(), abi.ProblemSet(), abi.ProblemSet()
        arrays.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], *sub, (0, 0, 0, 0),
                   cano5=fx["cano5"], cano3=fx["cano3"], dinc=dinc)
        p = codes.add(fx["a_codes"], fx["b_codes"], None, None, *sub, (0, 0, 0, 0))
        p.exin_left, p.exin_right = q["b_left"], q["b_right"]
        naive.add(fx["a_codes"], fx["b_codes"], None, None, *sub, (0, 0, 0, 0))
        sc_a = spdg.scoring(fx, scalar_engines=engines)
        sc_c = spdg.scoring(fx, scalar_engines=engines, sigmodel=model)
        want = eng.lsp_s(sc_a, arrays)[0]
        have = eng.lsp_s(sc_c, codes)[0]
        assert have[0] == want[0] and have[1].tolist() == want[1].tolist(), name
        other = eng.lsp_s(sc_c, naive)[0]
        n_diff += other[0] != want[0] or other[1].tolist() != want[1].tolist()
    eng.close()
