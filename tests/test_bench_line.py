"""bench.py's one stdout line stays small enough for the driver to parse (round 5's grew to 22 KB and was lost)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_compact_line_on_a_full_record():
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line_with_legs.json")))     # a real record with every leg
    assert len(json.dumps(full)) > 20000
    line = b.compact_line(full)
    assert "\n" not in line and len(line) <= b.LINE_LIMIT < 8192
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"]
    assert d["config"]["workload"].startswith("C2")
    rf = d["roofline"]
    assert rf["frac"] == full["roofline"]["frac"] and rf["bound"] == "hbm" and rf["peak"] == 8000.0
    assert rf["traffic"] == full["roofline"]["traffic"] and rf["kernel_ms"] > 0 and rf["valu"]["frac"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["cores"] == 16 and cb["value"] > 0
    legs = d["config"]["legs"]
    assert legs["a0"]["value"] == full["config"]["a0"]["value"]
    assert legs["c4_e2e"]["same"] == 9996 and legs["c4_e2e"]["of"] == 10000
    assert legs["dropin_q7_20k"]["Q7"]["x_ref"] > 1.0


def test_compact_line_never_exceeds_the_limit():
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line_with_legs.json")))
    for i in range(60):                                              # many more legs than exist: the line sheds them, keeps the contract
        full["config"][f"extra_{i}"] = dict(full["config"]["a0"])
    line = b.compact_line(full)
    assert len(line) <= b.LINE_LIMIT
    d = json.loads(line)
    assert d["roofline"]["frac"] and d["cpu_baseline"]["value"] and d["value"]


def test_the_round_record_is_within_the_limit():
    """profiles/r06_bench_legs.json (the full record of the round's last run) through the line builder, and the line kept beside it"""
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_legs.json")))
    line = b.compact_line(full)
    assert len(line) <= b.LINE_LIMIT
    d = json.loads(line)
    assert d["steps"] == 20 and d["warmup"] == 5 and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["kind"] == "reference"
    for leg in ("e2e_q7", "e2e_q7_p", "e2e_q7_s3", "c4_e2e"):
        assert d["config"]["legs"][leg]["same"] == d["config"]["legs"][leg]["of"], leg
    kept = open(os.path.join(ROOT, "profiles", "r06_bench_line.json")).read().strip()
    assert len(kept) <= b.LINE_LIMIT and json.loads(kept)["value"] == full["value"]
