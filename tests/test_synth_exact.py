"""synth.exact_inputs (cano5 / cano3 / dinc of a synthetic window, used by the -A0 workloads of bench.py) against the
arrays the reference itself built for the fixture windows (ref_dump: Exinon::isDonor / isAccpt, intron53_c classes)."""
import numpy as np
import pytest

from spaln_amd import synth
from tests import spdg
from tests.conftest import golden_files, golden_ids


@pytest.mark.parametrize("path", golden_files("s1_") + golden_files("c2_"), ids=golden_ids("s1_") + golden_ids("c2_"))
def test_exact_inputs_equal_the_references(path):
    fx = spdg.load(path)
    q = fx["prm"]
    if q["b_left"] != 0 or q["b_right"] != len(fx["b_codes"]):
        pytest.skip("sub-range fixture: the reference builds its arrays on the active range only")
    got = synth.exact_inputs(fx["b_codes"])
    n = len(fx["b_codes"])
    assert np.array_equal(got["dinc"], ((fx["dinc5"].astype(np.uint8) << 4) | fx["dinc3"].astype(np.uint8))[:n + 1])
    # the flags of the two positions at either end come out of the bytes in front of / behind the sequence in the
    # reference (the dinucleotide straddles the end): compared on the interior
    assert np.array_equal(got["cano5"][2:n - 1], (fx["cano5"] > 0).astype(np.uint8)[2:n - 1])
    assert np.array_equal(got["cano3"][2:n - 1], (fx["cano3"] > 0).astype(np.uint8)[2:n - 1])
