"""Probe: C2's batch as ONE resident batch on one context against TWO resident halves (split by DP cells, LPT) on two
contexts of the same device, each aligned by its own host thread -- with and without half a step of stagger -- so that
one half's ladder tail (forward sweeps, walks, SKL delivery) runs beside the other half's linear-space sweep.
    python tools/split_probe.py [--queries 10000] [--steps 5]
"""
import argparse, ctypes as C, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spaln_amd import abi, defaults, engine, shard, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=10000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--parts", type=int, default=2)
    a = ap.parse_args()
    sc = defaults.scoring()
    eng = engine.Engine(0)
    batch = synth.make_batch(a.queries, seed=synth.SEED)
    ps = abi.ProblemSet()
    for w, q, s5, s3, _ in batch:
        ps.add(q, w, s5, s3)
    costs = []
    for p in ps.items:
        win = abi.Window()
        eng.lib.spdp_stripe(C.byref(p), sc.sh, C.byref(win))
        costs.append(int(eng.lib.spdp_cells(C.byref(p), C.byref(win))))
    out = {}
    bt = eng.upload(sc, ps)
    for _ in range(2):
        bt.align(want=True, convert=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        _, _, kc = bt.align(want=True, convert=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    out["one_ms"] = round(dt * 1e3, 2); out["one_gcups"] = round(kc / dt / 1e9, 1)
    bt.free()
    parts = shard.balanced_shards(costs, a.parts)
    engs = [eng] + [engine.Engine(0) for _ in range(a.parts - 1)]
    bts = []
    for e, idx in zip(engs, parts):
        pp = abi.ProblemSet()
        for i in idx:
            w, q, s5, s3, _ = batch[i]
            pp.add(q, w, s5, s3)
        b = e.upload(sc, pp)
        b.align(want=True, convert=False); b.align(want=True, convert=False)
        bts.append(b)
    for stagger in (0.0, 0.5):
        cells = [0] * a.parts
        def run(k, delay):
            if delay: time.sleep(delay)
            for _ in range(a.steps):
                _, _, kc = bts[k].align(want=True, convert=False)
                cells[k] = kc
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ths = [threading.Thread(target=run, args=(k, stagger * k * dt / a.parts)) for k in range(a.parts)]
        for t in ths: t.start()
        for t in ths: t.join()
        torch.cuda.synchronize(); d2 = (time.perf_counter() - t0) / a.steps
        out[f"split{a.parts}_stagger{stagger}_ms"] = round(d2 * 1e3, 2)
        out[f"split{a.parts}_stagger{stagger}_gcups"] = round(sum(cells) / d2 / 1e9, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
