"""Double affine gaps (Noll = 3, the reference's -yl3) in the -A1 engines (round 5): scoreonlyS1 and forwardS1 as
spdp_exact<., ., DAGP> -- ev2 / fv2 beside ev / fv, five states a donor candidate can leave from, NCAND + 2 candidates per
lane (src/fwd2s1_simd.cc:347-455, 556-755; src/fwd2s1_simd.h:196-197, 270-274).  The reference's own HomScoreS_ng /
alignS_ng under `-yl3 -A1` (tests/golden/l3a1_*.spdg, traceback branch of the ladder), then random sub-ranges / end-gap
flags of the same pairs against the oracle, one group per problem and as a pipeline of stripes.  The linear-space engine
is refused: the reference's own hirschbergS1 is not usable under -yl3."""
import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files, golden_ids
from spaln_amd import abi, synth

pytestmark = pytest.mark.gpu

FILES = golden_files("l3a1_")


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("path", FILES, ids=golden_ids("l3a1_"))
def test_noll3_a1_equals_reference(eng, path):
    fx = spdg.load(path)
    assert fx["prm"]["noll"] == 3
    sc = spdg.scoring(fx, scalar_engines=2)
    ps, _ = spdg.problem(fx)
    assert int(eng.homscore_s(sc, ps)[0]) == int(fx["hom_scr_A1"][0])
    (scr, skl), = eng.align_s(sc, ps)
    assert scr == int(fx["aln_scr_A1"][0]) and skl.ravel().tolist() == fx["aln_skl_A1"].tolist()


def _subranges(fx, n, seed):
    q = fx["prm"]
    rng = np.random.default_rng(synth.SEED + seed)
    extra = dict(cano5=fx["cano5"], cano3=fx["cano3"],
                 dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
    ps = abi.ProblemSet()
    for i in range(n):
        m = int(rng.integers(min(12, q["a_right"]), q["a_right"] + 1))
        al = int(rng.integers(0, q["a_right"] - m + 1))
        bl = int(rng.integers(0, max(1, min(400, q["b_right"] - m - 200))))
        br = int(rng.integers(max(bl + m // 2 + 60, q["b_right"] - 600), q["b_right"] + 1))
        exg = (1, 1, 1, 1) if i % 3 == 0 else tuple(int(x) for x in rng.integers(0, 2, size=4))
        ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, al + m, bl, br, exg, **extra)
    return ps


@pytest.mark.parametrize("name", ["l3a1_long_gaps", "l3a1_divergent", "l3a1_local", "l3a1_900nt"])
def test_noll3_a1_subranges_against_oracle(eng, monkeypatch, name):
    from oracle import oracle, host_logic
    f = [f for f in FILES if f.endswith(name + ".spdg")]
    if not f:
        pytest.skip("fixture not present")
    fx = spdg.load(f[0])
    for local in ((0, 1) if name == "l3a1_local" else (0,)):
        sc = spdg.scoring(fx, scalar_engines=2, local=local, max_vmf_space=1 << 30)
        ps = _subranges(fx, 24, 900 + len(name))
        want_a = []
        for p in ps.items:
            try:
                want_a.append(host_logic.align_s(sc, p, simd=1))
            except (host_logic.NeedsScalarEngine, host_logic.ReferenceUndefined):
                want_a.append(None)
        assert sum(w is not None for w in want_a) >= 16
        for pipe in ("1", "0"):
            monkeypatch.setenv("SPDP_A1_PIPE", pipe)
            got_s = eng.homscore_s(sc, ps, allow_partial=True)
            res = eng.align_s(sc, ps, allow_partial=True)
            bad = []
            for i, (p, (score, skl), w) in enumerate(zip(ps.items, res, want_a)):
                if w is None:
                    continue
                if score != w[0] or skl.ravel().tolist() != (w[1] or []):
                    bad.append((local, pipe, i, (p.a_left, p.a_right, p.b_left, p.b_right), score, w[0]))
            assert not bad, bad[:4]
            # HomScoreS_ng: scoreonlyS1 on ranges of at least 4 rows (fewer go to the scalar engine, src/fwd2s1.cc:2700)
            for i, p in enumerate(ps.items):
                if p.a_right - p.a_left >= 4:
                    assert int(got_s[i]) == host_logic.homscore_s(sc, p, simd=1), (local, pipe, i)
        monkeypatch.delenv("SPDP_A1_PIPE")


def test_noll3_a1_linear_space_is_refused(eng):
    """hirschbergS1 under -yl3 has no defined result in the reference: the ladder must say so instead of running something"""
    f = [f for f in FILES if f.endswith("l3a1_900nt.spdg")][0]
    fx = spdg.load(f)
    sc = spdg.scoring(fx, scalar_engines=2, max_vmf_space=20000)       # too small for the traceback: the ladder wants hirschbergS1
    ps, _ = spdg.problem(fx)
    with pytest.raises(RuntimeError, match="hirschbergS1|linear-space"):
        eng.align_s(sc, ps)
