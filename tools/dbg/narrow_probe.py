"""Latency of the -A0 kernels on short-and-wide requests (what the seeded walk asks for across long introns):
30 query rows against the whole window of c5_6kb, one problem and 64 of them."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import spdg
from spaln_amd import abi, engine
fx = spdg.load("tests/golden/c5_6kb.spdg")
q = fx["prm"]
sc = spdg.scoring(fx, scalar_engines=1)
eng = engine.Engine(0)
extra = dict(cano5=fx["cano5"], cano3=fx["cano3"], dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
import sys as _s
if len(_s.argv) > 1 and _s.argv[1] == "nosites":            # no GT / AG anywhere: the step without donor / acceptor work
    extra["cano5"] = np.zeros_like(fx["cano5"]); extra["cano3"] = np.zeros_like(fx["cano3"])
for rows in (30, 100):
    for n in (1, 64):
        ps = abi.ProblemSet()
        for i in range(n):
            al = 500 + 40 * i
            ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, al + rows, 0, q["b_right"], (0, 0, 0, 0), **extra)
        for what, f in (("forward", lambda: eng.scalar_forward(sc, ps)), ("scorealone", lambda: eng.scalar_scorealone(sc, ps))):
            f()
            t = time.perf_counter(); f(); f(); dt = (time.perf_counter() - t) / 2
            if os.environ.get("SPDP_DBG_CNT"):                 # (a build with the event counters of DESIGN 6f; not in the shipped library)
                import ctypes as C
                c = (C.c_ulonglong * 8)(); eng.lib.spdp_dbg_counters(c, 1)
                print("   counters:", [int(x) >> 10 for x in list(c)[:7]])
            print(f"rows {rows} x cols {q['b_right']}, {n} problem(s), {what}: {dt * 1e3:.1f} ms = {dt / q['b_right'] * 1e6:.2f} us per column")
