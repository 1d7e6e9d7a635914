"""Map and align inside the library, end to end (tools/e2e_q7.py; round 5): the reference's index file and the genome in,
spdp_blk_find -> candidate loci -> spdp_align_s_seeded with the library's own HSP search -> spdp_skl_rng_s, and the exon
tables in chromosome coordinates against `spaln -Q7 -S1 -O4` of the compiled reference (oracle/_ref/spaln: test
infrastructure, prebuilt) on the same synthetic genome and queries."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exon_tables_equal_the_reference_program():
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "spaln")):
        pytest.skip("oracle/_ref/spaln is not built")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "e2e_q7.py"), "--queries", "300", "--genes", "60"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-400:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["reference_aligned"] == 300 and d["library_aligned"] == 300
    assert d["identical_exon_tables"] == 300, (d, r.stderr[-600:])
