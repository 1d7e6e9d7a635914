"""The oracle's restatement of the alignS_ng dispatch ladder (oracle/host_logic.py)
against the reference's own alignS_ng output (-A2 and -A3 engine selectors):
raw engine score and the final SKL corner list, bit-identical."""
import pytest

from tests import spdg
from tests.conftest import golden_files, golden_ids
from oracle import host_logic


@pytest.fixture(scope="module", params=golden_files(), ids=golden_ids())
def fx(request):
    return spdg.load(request.param)


@pytest.mark.parametrize("alg", [2, 3])
def test_align_s(fx, alg):
    sc = spdg.scoring(fx, nquant=(1 if alg == 3 else None))
    ps, p = spdg.problem(fx)
    try:
        scr, skl = host_logic.align_s(sc, p)
    except host_logic.NeedsScalarEngine:
        pytest.skip("needs the scalar engine (m < 8)")
    want = fx[f"aln_skl_A{alg}"].tolist()
    assert scr == int(fx[f"aln_scr_A{alg}"][0])
    assert (skl or []) == want
