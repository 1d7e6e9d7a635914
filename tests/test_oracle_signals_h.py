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


FILES = [f for pre in ("h1_", "c1_") for f in golden_files(pre) if "pm5_hdr" in spdg.load(f)]


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
