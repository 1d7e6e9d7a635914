"""The reference's own program on the library (integration/, built as oracle/_ref/spaln_gpu; test infrastructure like
oracle/_ref/spaln): `-O4` records against the unmodified build on a synthetic genome its own `spaln -W` formatted
(tools/dropin_demo.py), in the two orientation modes -- `-S1` (the queries as given) and spaln's default (`-S3`: both
orientations of every locus; every other query an antisense read) -- with and without seeding (-Q7 / -Q4)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("strand,extra", [("-S1", []), ("-S3", ["--antisense"]), ("-S2", ["--antisense"]), ("-S1", ["--protein"])],
                         ids=["S1", "S3_antisense", "S2_antisense", "protein"])
def test_records_identical_to_the_unmodified_program(strand, extra):
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "spaln_gpu")):
        pytest.skip("oracle/_ref/spaln_gpu is not built")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dropin_demo.py"), "--queries", "200", "--genes", "40", "--modes", "Q7,Q4",
                        "--gpu-threads", "16", f"--strand={strand}"] + extra, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-400:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for run in d["runs"]:
        if strand != "-S2":                             # (-S2 aligns the other orientation alone: the sense half of the reads finds little)
            assert run["reference"]["aligned"] == 200 and run["gpu"]["aligned"] == 200, run["mode"]
        else:
            assert run["reference"]["aligned"] == run["gpu"]["aligned"] >= 90, run["mode"]
        assert run["identical"] and run["records_differing"] == 0, (run["mode"], run["gpu"].get("shim", "")[:300])
        assert "left to the reference: 0" in run["gpu"].get("shim", ""), run["gpu"].get("shim", "")[:200]
