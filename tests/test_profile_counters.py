"""The counter profile bench.py prices its kernels with (profiles/rNN_counters.json, written by tools/profile_counters.py on
the GPU box) must belong to the kernels in the tree: every entry carries the sha256 of its kernel's source file.  A kernel
edited after its profile was taken fails here -- the figures in the bench line would be another kernel's."""
import hashlib
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _entries():
    import bench
    if not os.path.exists(bench.COUNTERS_FILE):
        pytest.skip("no counter profile committed yet for this round")
    with open(bench.COUNTERS_FILE) as f:
        allc = json.load(f)
    return [(wl, key, e) for wl, v in allc.items() for key, e in v["kernels"].items()]


def test_profile_entries_belong_to_the_kernels_in_the_tree():
    stale = []
    for wl, key, e in _entries():
        with open(os.path.join(ROOT, e["source"]), "rb") as f:
            if hashlib.sha256(f.read()).hexdigest() != e["source_sha256"]:
                stale.append(f"{wl}/{key}: {e['source']}")
    assert not stale, "kernel sources changed since the counter profile was taken (re-run tools/profile_counters.py): " + ", ".join(stale)


def test_profile_has_the_kernels_the_default_line_prices():
    have = {(wl, key) for wl, key, _ in _entries()}
    for need in [("c2", "udh"), ("c4", "forward"), ("c3", "h"), ("a0", "a0_udh")]:
        assert need in have, f"profile lacks {need}"
    for wl, key, e in _entries():
        assert e["launches_per_step"] > 0 and e["kernel"]
        if e.get("cells_per_step"):
            assert e.get("valu_per_cell", 0) > 0, (wl, key)
