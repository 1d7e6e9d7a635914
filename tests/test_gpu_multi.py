"""The N > 1 paths with the real engine, on one card: (a) a device group whose members are two contexts on GPU 0
shards a batch and returns exactly what one context returns, in query order; (b) bench.py --gpus 2 spawns two ranks
(BENCH_SHARE_GPU=1: both on GPU 0), reports n_gpus = 2 and both scaling modes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_group_equals_single_context():
    from spaln_amd import abi, defaults, engine, synth
    sc = defaults.scoring()
    ps = abi.ProblemSet()
    for w, q, s5, s3, _ in synth.make_batch(37, seed=2024, mrna_len=800, n_exons=5, flank=300, intron_hi=2500):
        ps.add(q, w, s5, s3)
    eng = engine.Engine(0)
    want_s = eng.homscore_s(sc, ps).tolist()
    want_a = [(s, skl.tolist()) for s, skl in eng.align_s(sc, ps)]
    eng.close()
    for members in ([0, 0], [0, 0, 0]):
        grp = engine.Group(members)
        assert grp.lib.spdp_group_size(grp.h) == len(members)
        assert grp.homscore_s(sc, ps).tolist() == want_s
        assert [(s, skl.tolist()) for s, skl in grp.align_s(sc, ps)] == want_a
        # the shards are balanced by DP cells (longest first to the least loaded member), not by count
        import ctypes as C
        from spaln_amd import shard
        member = (C.c_int32 * len(ps))()
        assert grp.lib.spdp_group_last_shards(grp.h, member, len(ps)) == len(ps)
        costs = []
        for p in ps.items:
            w = abi.Window()
            grp.lib.spdp_stripe(C.byref(p), sc.sh, C.byref(w))
            costs.append(int(grp.lib.spdp_cells(C.byref(p), C.byref(w))))
        want_sh = shard.balanced_shards(costs, len(members))
        assert [[i for i in range(len(ps)) if member[i] == r] for r in range(len(members))] == want_sh
        loads = [sum(costs[i] for i in s) for s in want_sh]
        assert max(loads) - min(loads) <= max(costs)
        grp.close()


@pytest.mark.timeout(900)
def test_bench_two_ranks_on_one_card():
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--queries", "600", "--cpu-sample", "4"], capture_output=True, text=True, env=env, timeout=850)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["queries_total"] == 1200
    strong = rec["config"]["strong_scaling"]
    assert strong["queries_total"] == 600 and strong["value"] > 0
    assert len(strong["rank_busy_ms"]) == 2 and len(strong["rank_cells"]) == 2
    assert max(strong["rank_cells"]) - min(strong["rank_cells"]) < 0.02 * sum(strong["rank_cells"])   # cell-balanced shards
    assert rec["cpu_baseline"] is None                 # timed at N = 1 only
