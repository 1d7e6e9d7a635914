"""SpJunc::spjseq (src/codepot.cc:79-107) with ambiguous bases, against the reference's own function: the two h1_amb_junction
fixtures carry N two before donors / one behind acceptors, and `ref_dump` records spjseq(n5, n3) for pairs around every
ambiguous position (rows {n5, n3, phase-1 codon, phase-2 codon}).  The rule all restatements follow: a codon is defined
when its own three bases are (the reference's spj_amb_tron_tab / spj_tron_amb_tab)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle, seeded, host_logic_h as hh
from tests import spdg
from tests.conftest import golden_files

FILES = golden_files("h1_amb_junction")


@pytest.mark.parametrize("path", FILES, ids=[f.split("/")[-1][:-5] for f in FILES])
def test_split_codons_equal_reference(path):
    fx = spdg.load(path)
    _, p = spdg.problem_h(fx)
    probe = np.asarray(fx["spj_probe"]).reshape(-1, 4)
    assert len(probe) > 50
    kinds = set()
    b = fx["b_codes"]
    cs = (C.c_int32 * 2)()
    for n5, n3, c0, c1 in probe.tolist():
        oracle.lib().orc_spjseq_h(C.byref(p), C.c_int(n5), C.c_int(n3), cs)
        assert (cs[0], cs[1]) == (c0, c1), (n5, n3)
        assert hh._spjseq(b, p.b_left, p.b_right, n5, n3) == (c0, c1), (n5, n3)
        assert seeded.lib().walk_check_split_codon_h(C.byref(p), C.c_int(n5), C.c_int(n3), cs) == 0
        assert (cs[0], cs[1]) == (c0, c1), ("walk", n5, n3)
        kinds.add((c0 == 2, c1 == 2))
    # all four outcomes occur: both codons, only the first, only the second, neither
    assert kinds == {(False, False), (False, True), (True, False), (True, True)}
