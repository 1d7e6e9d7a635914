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


def test_eight_member_group_balances_cells_and_keeps_query_order():
    """eight contexts on the one card stand in for the eight GPUs of a node: shards by DP cells within 2 %, results in
    the caller's order"""
    from spaln_amd import abi, defaults, engine, synth
    import ctypes as C
    sc = defaults.scoring()
    ps = abi.ProblemSet()
    for w, q, s5, s3, _ in synth.make_batch(400, seed=77, mrna_len=600, n_exons=4, flank=200, intron_hi=1500):
        ps.add(q, w, s5, s3)
    eng = engine.Engine(0)
    want = eng.homscore_s(sc, ps).tolist()
    eng.close()
    grp = engine.Group([0] * 8)
    assert grp.homscore_s(sc, ps).tolist() == want
    member = grp.shards(len(ps))
    costs = []
    for p in ps.items:
        w = abi.Window()
        grp.lib.spdp_stripe(C.byref(p), sc.sh, C.byref(w))
        costs.append(int(grp.lib.spdp_cells(C.byref(p), C.byref(w))))
    loads = [sum(c for c, m in zip(costs, member) if m == r) for r in range(8)]
    assert min(loads) > 0 and (max(loads) - min(loads)) / max(loads) < 0.02, loads
    grp.close()


def test_group_seeded_calls_equal_the_reference():
    """spdp_group_align_s_seeded / _h_seeded: a fixture group with one parameter set through three members; the HSP source is
    asked with the caller's query numbers (every query has its own recorded Wilip replies), results = the reference's"""
    from spaln_amd import abi, engine
    from oracle import seeded
    from tests import spdg
    from tests.test_oracle_seeded import Q_FILES
    from tests.test_oracle_seeded_h import QH, seeded_inputs_h, UNDEFINED
    grp = engine.Group([0, 0, 0])
    groups = {}
    for f in Q_FILES:
        fx = spdg.load(f)
        seedp = [int(x) for x in fx["seed_params"]]
        key = (seedp[0], tuple(seedp[3:]), tuple(int(x) for x in fx["params"][:19]), int(fx["params"][27]), int(fx["params"][28]))
        groups.setdefault(key, []).append(fx)
    fxs = max(groups.values(), key=len)
    assert len(fxs) >= 4
    ps = abi.ProblemSet()
    hs, lv, wls, keep = [], [], [], []
    for fx in fxs:
        spdg.problem(fx, ps)
        p = ps.items[-1]
        h5, h3 = np.ascontiguousarray(fx["phs5"]), np.ascontiguousarray(fx["phs3"])
        keep += [h5, h3]
        p.phs5, p.phs3 = h5.ctypes.data, h3.ctypes.data
        sp = abi.seed_params_from_fixture(fx)
        j, n = seeded.hsps_of(fx)
        hs.append(j if n else None)
        lv.append(int(fx["seed_params"][1]))
        wls.append(seeded.parse_wilip_log(fx["seed_wilip_A2"]))
    sc = spdg.scoring(max(fxs, key=lambda f: len(f["intpen"])))
    res = grp.align_s_seeded(sc, sp, ps, hs, lv, wls)
    assert len(set(grp.shards(len(fxs)).tolist())) > 1
    for fx, (scr, skl) in zip(fxs, res):
        assert scr == int(fx["seed_scr_A2"][0]) and [int(x) for x in skl.ravel()] == fx["seed_skl_A2"].tolist()
    # protein: the fixtures one by one through the group entry (each its own parameter set)
    n_ok = 0
    for path in QH[:8]:
        name = path.split("/")[-1][:-5]
        if (name, 2) in UNDEFINED:
            continue
        fx = spdg.load(path)
        sch, sph, p, hsps, n, lowest, wl = seeded_inputs_h(fx, 2)
        sch.scalar_engines = 0
        (scr, skl), = grp.align_h_seeded(sch, sph, p._owner, [hsps if n else None], [lowest], [wl])
        assert scr == int(fx["seed_scr_A2"][0]) and [int(x) for x in skl.ravel()] == fx["seed_skl_A2"].tolist()
        n_ok += 1
    assert n_ok >= 5
    grp.close()


def test_group_block_vote_equals_recorded_runs():
    import ctypes as C
    from spaln_amd import blocks, engine
    from oracle import blk as oblk
    from tests import spdg
    from tests import test_gpu_blk as tb
    fx = spdg.load(os.path.join(ROOT, "tests", "golden", "blk_k1.spdg"))
    qs = oblk.parse_log(fx)
    grp = engine.Group([0, 0, 0])
    lib = grp.lib
    lib.spdp_group_context.restype = C.c_void_p
    lib.spdp_group_context.argtypes = [C.c_void_p, C.c_int]

    class Member:                                   # what blocks.BlockIndex needs of an engine
        def __init__(self, ctx):
            self.lib, self.ctx = lib, ctx

        def _check(self, rc, what):
            assert rc == 0, what
    idx = [blocks.BlockIndex(Member(lib.spdp_group_context(grp.h, r)), fx) for r in range(3)]
    handles = (C.c_void_p * 3)(*[i.h for i in idx])
    queries = [q["codes"] for q in qs]
    offs = np.zeros(len(queries) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(q) for q in queries])
    codes = np.ascontiguousarray(np.concatenate(queries))
    left = np.array([q["left"] for q in qs], dtype=np.int32)
    right = np.array([q["right"] for q in qs], dtype=np.int32)
    cap = 1 << 14
    out = np.zeros((len(qs), cap), dtype=np.int32)
    lib.spdp_group_blk_vote.argtypes = [C.c_void_p] * 7 + [C.c_int32, C.c_void_p, C.c_int32]
    rc = lib.spdp_group_blk_vote(grp.h, handles, codes.ctypes.data, offs.ctypes.data, left.ctypes.data, right.ctypes.data, None,
                                 len(qs), out.ctypes.data, cap)
    assert rc == 0, lib.spdp_group_last_error(grp.h)
    tb._IX[0] = oblk.index_of(fx)[0]
    _keep = oblk.index_of(fx)
    tb._IX[0] = _keep[0]
    for i, q in enumerate(qs):
        assert tb.same(blocks.split_record(out[i]), oblk.split_recorded(*q["calls"][0])), i
    assert len(set(grp.shards(len(qs)).tolist())) == 3
    for i in idx:
        i.free()
    grp.close()


def test_group_map_align_equals_one_context():
    """spdp_group_map_align_s (queries sharded over three members, an index per member) against spdp_map_align_s on one
    context: the same genes and exon tables in the caller's order, both orientations tried"""
    import ctypes as C
    from spaln_amd import abi, blocks, engine
    from oracle import blk as oblk
    from tests import spdg
    from tests.test_blk_find import genome_of
    fx = spdg.load(os.path.join(ROOT, "tests", "golden", "blk_par.spdg"))
    fq = spdg.load(os.path.join(ROOT, "tests", "golden", "q_c2_seed0.spdg"))
    gen, off = genome_of("blk_par", 24, 980, True)
    queries = [q["codes"][q["left"]:q["right"]] for q in oblk.parse_log(fx)]
    comp = np.arange(256, dtype=np.uint8)
    for a, b in ((2, 9), (9, 2), (3, 5), (5, 3)):
        comp[a] = b
    queries = [comp[q[::-1]] if i % 2 else q for i, q in enumerate(queries)]        # antisense reads among them
    model = abi.wilip_model_from_fixture(fq)
    sigmodel = abi.signal_model_from_fixture(fq)
    prm = blocks.find_params_from_fixture(fx)
    sc = spdg.scoring(fq, intpen=np.ascontiguousarray(fx["find_intpen"], dtype=np.int16), scalar_engines=1)
    sp = abi.seed_params_from_fixture(fq)
    sp.wilip = C.addressof(model)
    fs = fq["rng_fstat_A0"] if "rng_fstat_A0" in fq else [0, 0, 0, 0, 0, 0, 3, 1]
    rescore = (fq["prm"]["codonk1"], fq["prm"]["minl"], int(fs[6]), int(fs[7]))
    eng = engine.Engine(0)
    dix = blocks.BlockIndex(eng, fx)
    want, _, rc = blocks.map_align(dix, gen, off, sc, sp, sigmodel, prm, rescore, queries, ori=3)
    assert rc == 0 and sum(1 for g in want if g is not None) >= 10 and any(g and g["q_rev"] for g in want)
    dix.free()
    eng.close()

    grp = engine.Group([0, 0, 0])
    lib = grp.lib
    lib.spdp_group_context.restype = C.c_void_p
    lib.spdp_group_context.argtypes = [C.c_void_p, C.c_int]

    class Member:
        def __init__(self, ctx):
            self.lib, self.ctx = lib, ctx

        def _check(self, rc, what):
            assert rc == 0, what
    idx = [blocks.BlockIndex(Member(lib.spdp_group_context(grp.h, r)), fx) for r in range(3)]
    handles = (C.c_void_p * 3)(*[i.h for i in idx])
    n = len(queries)
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(q) for q in queries])
    codes = np.ascontiguousarray(np.concatenate(queries))
    g = blocks.Genome()
    gc = np.ascontiguousarray(gen, dtype=np.uint8)
    go = np.ascontiguousarray(off, dtype=np.int64)
    g.codes, g.chr_off, g.n_chr = gc.ctypes.data, go.ctypes.data, len(go) - 1
    rp = abi.RescoreParams(*(int(x) for x in rescore))
    genes = (blocks.MapGene * n)()
    exons = C.POINTER(blocks.MapExon)()
    lib.spdp_group_map_align_s.restype = C.c_int
    lib.spdp_group_map_align_s.argtypes = [C.c_void_p] * 11 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    rc = lib.spdp_group_map_align_s(grp.h, handles, C.byref(idx[0].desc), C.byref(g), C.byref(sc), C.byref(sp), C.addressof(sigmodel),
                                    C.byref(prm), C.byref(rp), codes.ctypes.data, offs.ctypes.data, n, 3, genes, C.byref(exons))
    assert rc == 0, lib.spdp_group_last_error(grp.h)
    for i in range(n):
        G = genes[i]
        if want[i] is None:
            assert G.chr < 0, i
            continue
        ex = [(exons[G.exon_off + j].q_left, exons[G.exon_off + j].q_right, exons[G.exon_off + j].g_left, exons[G.exon_off + j].g_right)
              for j in range(G.n_exons)]
        assert (G.chr, G.rvs, G.q_rev, G.score, G.val, G.n_loci, ex) == tuple(want[i][k] for k in ("chr", "rvs", "q_rev", "score", "val", "n_loci", "exons")), i
    assert len(set(grp.shards(n).tolist())) == 3
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(exons)
    for i in idx:
        i.free()
    grp.close()
