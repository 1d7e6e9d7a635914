"""SpdpProblemH.cip on the protein path (Cip_score::cip_score(3 m - phs), src/fwd2h1.cc:352-354, 483; the -A1 form
fwd2h1_simd.h:407): GPU -A0 and -A1 ladders against the oracle's restatement with random sparse bonuses."""
import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("engines", [1, 2])
def test_cip_h_against_oracle(engines):
    from spaln_amd import abi, engine
    from oracle import host_logic_h as hh
    rng = np.random.default_rng(501 + engines)
    eng = engine.Engine(0)
    changed = n_cmp = 0
    for name in ("h1_basic", "h1_400aa", "h1_frameshift", "h1_divergent"):
        fx = spdg.load([f for f in golden_files("h1_") if f.endswith(name + ".spdg")][0])
        sc = spdg.scoring_h(fx, scalar_engines=engines)
        plain, bonus = abi.ProblemSetH(), abi.ProblemSetH()
        _, p0 = spdg.problem_h(fx, plain)
        cip = np.zeros(3 * fx["a_codes"].size + 2, dtype=np.int32)
        hit = rng.random(cip.size) < 0.4
        cip[hit] = rng.integers(40, 300, size=int(hit.sum()))
        _, p1 = spdg.problem_h(fx, bonus)
        keep = np.ascontiguousarray(cip)
        bonus._keep.append(keep)
        p1.cip = keep.ctypes.data
        bonus.items[-1] = p1
        r0 = eng.align_h(sc, plain)[0]
        r1 = eng.align_h(sc, bonus)[0]
        try:
            ws, wskl = hh.align_h(sc, p1, simd=0 if engines == 1 else 1)
        except (hh.ReferenceUndefined, hh.ReferenceFatal):
            continue
        n_cmp += 1
        assert r1[2] == 0 and r1[0] == ws and r1[1].ravel().tolist() == (wskl or []), (engines, name)
        changed += r1[0] != r0[0] or r1[1].tolist() != r0[1].tolist()
    eng.close()
    # (under -A1 alignH_ng reports nevsel as its score, as the reference does: only the records can move there)
    assert n_cmp >= 3 and (changed >= 2 or engines == 2)
