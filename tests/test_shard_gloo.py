"""N > 1 path on CPU: two gloo processes shard a query list, "align" their slice (with the
oracle standing in for the GPU engine -- this test checks the sharding / gather logic, not
the kernels) and the gathered result equals the single-process result, in query order."""
import os
import subprocess
import sys
import textwrap

import pytest

from spaln_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for n in (0, 1, 7, 10, 10000):
        for w in (1, 2, 3, 8):
            seen = []
            for r in range(w):
                seen += list(shard.shard_range(n, r, w))
            assert seen == list(range(n))
            sizes = [len(shard.shard_range(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def test_balanced_shards_partition_and_balance():
    import numpy as np
    rng = np.random.default_rng(5)
    for n in (0, 1, 7, 500):
        costs = rng.integers(6_000_000, 33_000_000, size=n).tolist()     # 2 kb x 3-16 kb windows
        for w in (1, 2, 3, 8):
            sh = shard.balanced_shards(costs, w)
            assert sorted(i for s in sh for i in s) == list(range(n)) and all(s == sorted(s) for s in sh)
            if n >= 100:
                loads = [sum(costs[i] for i in s) for s in sh]
                by_count = [sum(costs[i] for i in shard.shard_range(n, r, w)) for r in range(w)]
                assert max(loads) - min(loads) <= max(costs)                # LPT: within one item of each other
                assert max(loads) <= max(by_count)
            parts = [[("r", i) for i in s] for s in sh]
            assert shard.scatter_in_order(n, sh, parts) == [("r", i) for i in range(n)]


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from spaln_amd import abi, defaults, shard, synth
    from oracle import oracle
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    sc = defaults.scoring()
    batch = synth.make_batch(9, seed=99, n_exons=3, mrna_len=200, flank=80, intron_hi=300)
    costs = []
    for w, q, s5, s3, _ in batch:
        ps = abi.ProblemSet(); p = ps.add(q, w, s5, s3)
        costs.append(oracle.cells(p, oracle.stripe(p, sc.sh)))
    shards = shard.balanced_shards(costs, world)          # by DP cells, the same list on every rank
    mine = shards[rank]
    local = []
    for i in mine:
        w, q, s5, s3, _ = batch[i]
        ps = abi.ProblemSet(); p = ps.add(q, w, s5, s3)
        local.append((i, oracle.wip_scoreonly(sc, p)))
    parts = [None] * world
    dist.all_gather_object(parts, local)
    full = shard.scatter_in_order(len(batch), shards, parts)
    if rank == 0:
        print("RESULT " + json.dumps(full))
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 8])              # 8: the ranks of a full node (more ranks than some have work for)
def test_gloo_ranks_match_single(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(29641 + world), str(script)],
                         capture_output=True, text=True, env=env, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    import json
    got = json.loads(line[7:])
    # single-process reference
    from spaln_amd import abi, defaults, synth
    from oracle import oracle
    sc = defaults.scoring()
    batch = synth.make_batch(9, seed=99, n_exons=3, mrna_len=200, flank=80, intron_hi=300)
    want = []
    for i, (w, q, s5, s3, _) in enumerate(batch):
        ps = abi.ProblemSet(); p = ps.add(q, w, s5, s3)
        want.append([i, oracle.wip_scoreonly(sc, p)])
    assert got == want
