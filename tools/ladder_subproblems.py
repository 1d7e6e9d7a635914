"""which hirschbergS1 call of a ladder differs: every sub-problem the oracle's -A1 ladder hands to the linear-space engine is
run as a problem of its own on the GPU and in the oracle (GPU box: python tools/ladder_subproblems.py fixture al ar bl br exg vmf)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import spdg
from tests.conftest import golden_files
from spaln_amd import abi, engine
from oracle import host_logic, oracle

name, al, ar, bl, br, exg, vmf = sys.argv[1], *[int(x) for x in sys.argv[2:6]], sys.argv[6], int(sys.argv[7])
fx = spdg.load([f for f in golden_files("s1_") if f.endswith(name + ".spdg")][0])
extra = dict(cano5=fx["cano5"], cano3=fx["cano3"], dinc=(fx["dinc5"].astype("uint8") << 4) | fx["dinc3"].astype("uint8"))
sc = spdg.scoring(fx, scalar_engines=2, max_vmf_space=vmf)
calls = []
real = oracle.exact_udh


def spy(sc_, p, n_imd, w=None):
    out = real(sc_, p, n_imd, w)
    calls.append(((p.a_left, p.a_right, p.b_left, p.b_right), (p.a_exgl, p.a_exgr, p.b_exgl, p.b_exgr), n_imd,
                  int(out[0]), [int(x) for x in out[1][0][:4]], [int(x) for x in out[2]]))
    return out


oracle.exact_udh = spy
ps = abi.ProblemSet()
p = ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], al, ar, bl, br, tuple(int(c) for c in exg), **extra)
host_logic.align_s(sc, p, simd=1)
oracle.exact_udh = real
eng = engine.Engine(0)
for rng, flags, n_imd, scr, c0, wr in calls:
    ps = abi.ProblemSet()
    q = ps.add(fx["a_codes"], fx["b_codes"], fx["sig5"], fx["sig3"], *rng, flags, **extra)
    (score, skl), = eng.align_s(sc, ps, allow_partial=True)
    try:
        ws, wskl = host_logic.align_s(sc, q, simd=1)
    except Exception as e:
        ws, wskl = repr(e), []
    same = score == ws and skl.ravel().tolist() == (wskl or [])
    print(rng, "exg", flags, "n_imd", n_imd, "oracle udh:", scr, c0, wr, "| as a problem of its own: GPU", score, len(skl), "oracle", ws,
          len(wskl or []) // 2, "SAME" if same else "DIFFERENT", flush=True)
eng.close()
