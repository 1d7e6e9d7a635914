"""BASELINE config 1 (dictdisc proteins, species tables -Tdictdisc) on the GPU: the c1_* fixtures through the `_wip`
engines and the alignH_ng ladder under -A2 / -A3 (the -A0 / -A1 engines run them in test_gpu_scalar_h / _exact_h)."""
import pytest

from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi

pytestmark = pytest.mark.gpu
C1 = golden_files("c1_")


def _name(f):
    return f.split("/")[-1][:-5]


@pytest.fixture(scope="module")
def eng():
    from spaln_amd import engine
    e = engine.Engine(0)
    yield e
    e.close()


def test_c1_fixtures_exist():
    assert len(C1) >= 6


@pytest.mark.parametrize("tag", ["qn", "q1"])
def test_c1_forward_wip(eng, tag):
    cases = [(_name(f), spdg.load(f)) for f in C1]
    sc = spdg.scoring_h(cases[0][1], nquant=None if tag == "qn" else 1)
    ps = abi.ProblemSetH()
    for _, fx in cases:
        spdg.problem_h(fx, ps)
    res = eng.wip_forward_h(sc, ps)
    for (name, fx), (score, skl, flag) in zip(cases, res):
        assert flag == 0 and score == int(fx[f"wip_{tag}_fwd_scr"][0]), name
        assert skl.ravel().tolist() == fx[f"wip_{tag}_fwd_skl"].tolist(), name


@pytest.mark.parametrize("alg", [2, 3])
def test_c1_align_h(eng, alg):
    cases = [(_name(f), spdg.load(f)) for f in C1]
    sc = spdg.scoring_h(cases[0][1], nquant=None if alg == 2 else 1)
    ps = abi.ProblemSetH()
    for _, fx in cases:
        spdg.problem_h(fx, ps)
    res = eng.align_h(sc, ps)
    hom = eng.homscore_h(sc, ps)
    for (name, fx), (score, skl, flag), hs in zip(cases, res, hom):
        assert flag == 0 and score == int(fx[f"aln_scr_A{alg}"][0]), name
        assert skl.ravel().tolist() == fx[f"aln_skl_A{alg}"].tolist(), name
        assert int(hs) == int(fx[f"hom_scr_A{alg}"][0]), name
        assert len(skl) >= 4, name                     # a spliced alignment came out
