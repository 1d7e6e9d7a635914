"""spdp_align_s_ori3 against the reference's own alignS_ng(seqs, pwd, gsi, 3) output (fixtures o3_*, ref_dump -O):
orientation picked by infer_orientation, gsi->scr, the SKL with the A_RevCom bit -- under -A2 (the `_wip` engines)
and under -A0 (scalar_engines = 1)."""
import pytest

from tests import spdg
from tests.conftest import golden_files, golden_ids

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("alg", [2, 0])
def test_align_s_ori3_vs_reference(alg):
    from spaln_amd import abi, engine
    eng = engine.Engine(0)
    n = 0
    for path in golden_files("o3_"):
        fx = spdg.load(path)
        sc = spdg.scoring(fx, scalar_engines=1 if alg == 0 else 0)
        fwd, _ = spdg.problem(fx)
        rev, _ = spdg.problem_rev(fx)
        res, orient = eng.align_s_ori3(sc, fwd, rev)
        assert int(orient[0]) == int(fx[f"ori3_rev_A{alg}"][0]), path
        assert res[0][0] == int(fx[f"ori3_scr_A{alg}"][0]), path
        assert res[0][1].ravel().tolist() == fx[f"ori3_skl_A{alg}"].tolist(), path
        n += 1
    eng.close()
    assert n >= 5
