"""SURVEY 8 f1, protein side: the CPU restatement of Exinon::intron53_p (oracle/signals_h.py: position weight matrices of
order 1 and 2 on the tron sequence, the 5th-order coding potential, stop-codon rules, dinucleotide classes, phases)
against the SGPT6 arrays the reference produced for every protein fixture -- h1_* with the generic tables, c1_* with the
Dictyostelium set: all seven arrays, every position."""
import numpy as np
import pytest

from oracle import signals_h
from tests import spdg
from tests.conftest import golden_files

def stale_mask(b_len, left, right):
    """the cells the reference's class loop never assigns (acceptor class of `left`, donor class of `right - 1`: whatever
    the allocation held, as on the cDNA path) and the phases derived from them"""
    ok = {k: np.ones(b_len + 3, dtype=bool) for k in ("sig5", "sig3", "phs5", "phs3")}
    ok["sig3"][left] = False
    ok["phs3"][max(left - 1, 0):left + 2] = False
    ok["sig5"][right - 1] = False
    ok["phs5"][max(right - 2, 0):right + 1] = False
    return ok


FILES = [f for pre in ("h1_", "c1_", "hb_") for f in golden_files(pre) if "pm5_hdr" in spdg.load(f)]   # hb_: branch-point term on (own intron-penalty table: not batched with h1_)


@pytest.mark.parametrize("path", FILES, ids=[f.split("/")[-1][:-5] for f in FILES])
def test_signals_h_equal_reference(path):
    fx = spdg.load(path)
    md = signals_h.model_of(fx)
    q = fx["prm"]
    b_len = len(fx["b_codes"]) - 1
    got = signals_h.splice_signals_h(md, fx["b_codes"], b_len, q["b_left"], q["b_right"])
    ok = stale_mask(b_len, q["b_left"], q["b_right"])
    for k in ("sig5", "sig3", "sigS", "sigT", "sigE", "phs5", "phs3"):
        m = ok[k] if k in ok else np.ones(b_len + 3, dtype=bool)
        assert np.array_equal(got[k][m], fx[k][:b_len + 3][m]), k


def test_fixture_count():
    assert len(FILES) >= 34


def test_branch_point_term_is_exercised():
    """hb_branch*: `ref_dump -b` (-yB) switches the branch-point matrix on; the acceptor signal then differs from the plain one
    at hundreds of positions, and the reach limit (-yD) matters"""
    fa, fb = (spdg.load([f for f in golden_files("hb_branch") if f.endswith(n + ".spdg")][0]) for n in ("hb_branch", "hb_branch_d20"))
    md = signals_h.model_of(fa)
    assert md["pmB"].present and md["fB"] > 0
    q = fa["prm"]
    b_len = len(fa["b_codes"]) - 1
    with_b = signals_h.splice_signals_h(md, fa["b_codes"], b_len, max(0, q["b_left"]), q["b_right"])["sig3"]
    md0 = dict(md)
    md0["pmB"] = signals_h.PatMat([0, 0, 0, 0, 0], fa["pmB_f32"][:2])
    without = signals_h.splice_signals_h(md0, fa["b_codes"], b_len, max(0, q["b_left"]), q["b_right"])["sig3"]
    assert int(np.count_nonzero(with_b != without)) > 200
    assert int(np.count_nonzero(np.asarray(fa["sig3"]) != np.asarray(fb["sig3"]))) > 100
