"""one sub-range of a cDNA fixture through alignS_ng under -A1 / -A0 on the GPU and in the oracle (GPU box:
python tools/ladder_case.py fixture a_left a_right b_left b_right exg(4 digits) max_vmf_space)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi, engine
from oracle import host_logic

name, al, ar, bl, br, exg, vmf = sys.argv[1], *[int(x) for x in sys.argv[2:6]], sys.argv[6], int(sys.argv[7])
fx = spdg.load([f for f in golden_files("s1_") if f.endswith(name + ".spdg")][0])
extra = dict(cano5=fx["cano5"], cano3=fx["cano3"], dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
eng = engine.Engine(0)
for sel, simd in ((2, 1), (1, 0)):
    for v in (vmf, 1 << 30):
        sc = spdg.scoring(fx, scalar_engines=sel, max_vmf_space=v)
        ps = abi.ProblemSet()
        p = ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, ar, bl, br, tuple(int(c) for c in exg), **extra)
        (score, skl), = eng.align_s(sc, ps, allow_partial=True)
        try:
            ws, wskl = host_logic.align_s(sc, p, simd=simd)
        except Exception as e:
            ws, wskl = repr(e), []
        g = skl.ravel().tolist()
        w = wskl or []
        first = next((i for i in range(min(len(g), len(w))) if g[i] != w[i]), min(len(g), len(w)))
        print(f"-A{1 if simd else 0} MaxVmfSpace {v}: GPU {score} ({len(g) // 2 - 1} corners)  oracle {ws} ({len(w) // 2 - 1} corners)  "
              f"equal {g == w}  first difference at {first}: {g[max(0, first - 4):first + 8]} | {w[max(0, first - 4):first + 8]}", flush=True)
        if g != w and os.environ.get("FULL"):
            print("  GPU   ", g)
            print("  oracle", w)
eng.close()
