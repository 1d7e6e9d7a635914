"""SURVEY 8 f1, protein side, on the device: spdp_splice_signals_h (spdp_signals_h.hip) against the reference's SGPT6
arrays in the protein fixtures and against the oracle; a batch uploaded as tron codes only (SpdpScoringH::sigmodel)
against the same batch with host-supplied arrays."""
import numpy as np
import pytest

from tests import spdg
from tests.conftest import golden_files
from tests.test_oracle_signals_h import FILES, stale_mask

pytestmark = pytest.mark.gpu
KEYS = ("sig5", "sig3", "sigS", "sigT", "sigE", "phs5", "phs3")


@pytest.mark.parametrize("path", FILES, ids=[f.split("/")[-1][:-5] for f in FILES])
def test_device_signals_h_equal_reference(path):
    from spaln_amd import abi, engine
    from oracle import signals_h
    fx = spdg.load(path)
    model = abi.signal_model_h_from_fixture(fx)
    q = fx["prm"]
    b_len = len(fx["b_codes"]) - 1
    eng = engine.Engine(0)
    got = eng.splice_signals_h(model, fx["b_codes"], q["b_left"], q["b_right"])
    eng.close()
    ok = stale_mask(b_len, q["b_left"], q["b_right"])
    for k in KEYS:
        m = ok[k] if k in ok else np.ones(b_len + 3, dtype=bool)
        assert np.array_equal(got[k][m], fx[k][:b_len + 3][m]), k
    # and every cell, stale ones included, against the oracle (class 0 there)
    want = signals_h.splice_signals_h(signals_h.model_of(fx), fx["b_codes"], b_len, q["b_left"], q["b_right"])
    for k in KEYS:
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["dinc"][:b_len + 1] >> 4, fx["dinc5"][:b_len + 1]) and \
        np.array_equal(got["dinc"][:b_len + 1] & 15, fx["dinc3"][:b_len + 1])


@pytest.mark.parametrize("engines", [0, 1])
def test_batch_from_tron_codes_equals_reference(engines):
    """the reference's own alignment from tron codes + model alone (no signal array crosses PCIe): -A2 and -A0 ladders,
    generic and dictdisc tables"""
    from spaln_amd import abi, engine
    eng = engine.Engine(0)
    n = 0
    for path in FILES:
        fx = spdg.load(path)
        name = path.split("/")[-1][:-5]
        q = fx["prm"]
        alg = 2 if engines == 0 else 0
        if f"aln_scr_A{alg}" not in fx or q["local"] or name in ("h1_random", "h1_cut_right"):
            continue
        model = abi.signal_model_h_from_fixture(fx)
        b_len = len(fx["b_codes"]) - 1
        got = eng.splice_signals_h(model, fx["b_codes"], q["b_left"], q["b_right"])
        if not all(np.array_equal(got[k], fx[k][:b_len + 3]) for k in KEYS):
            continue                                        # a stale boundary cell in the reference's run
        ps = abi.ProblemSetH()
        ps.add(fx["a_codes"], fx["b_codes"], None, None, None, None, None, None, None, q["a_left"], q["a_right"],
               q["b_left"], q["b_right"], (q["a_exgl"], q["a_exgr"], q["b_exgl"], q["b_exgr"]))
        sc = spdg.scoring_h(fx, scalar_engines=engines, sigmodel=model)
        (score, skl, flag), = eng.align_h(sc, ps)
        assert flag == 0 and score == int(fx[f"aln_scr_A{alg}"][0]), (name, engines)
        assert skl.ravel().tolist() == fx[f"aln_skl_A{alg}"].tolist(), (name, engines)
        n += 1
    eng.close()
    assert n >= 8
