"""Diagnostic sweep for the GPU box: runs every golden through the three HIP
engines and writes a one-line verdict per case to gpurun_out/diag.txt."""
import os
import re
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import spdg  # noqa: E402
from tests.conftest import golden_files  # noqa: E402
from spaln_amd import engine  # noqa: E402


def main():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    out = open(os.path.join(ROOT, "gpurun_out", "diag.txt"), "w")
    eng = engine.Engine(0)
    print(eng.device_name(), file=out)
    for f in golden_files():
        name = os.path.basename(f)[:-5]
        fx = spdg.load(f)
        for tag in ("qn", "q1"):
            sc = spdg.scoring(fx, nquant=(1 if tag == "q1" else None))
            ps, p = spdg.problem(fx)
            line = [name, tag]
            try:
                s = int(eng.wip_scoreonly(sc, ps)[0])
                w = int(fx["wip_%s_score" % tag][0])
                line.append("score OK" if s == w else "score BAD %d != %d" % (s, w))
            except Exception as e:
                line.append(f"score EXC {e}")
            try:
                (s, skl), = eng.wip_forward(sc, ps)
                ws = int(fx["wip_%s_fwd_scr" % tag][0])
                wk = fx["wip_%s_fwd_skl" % tag].tolist()
                ok = s == ws and skl.ravel().tolist() == wk
                line.append("fwd OK" if ok else "fwd BAD scr %d vs %d skl %s vs %s" % (s, ws, skl.ravel().tolist()[:12], wk[:12]))
            except Exception as e:
                line.append(f"fwd EXC {e}")
            if not sc.local:
                for k in [k for k in fx if re.fullmatch(rf"wip_{tag}_udh\d+_scr", k)]:
                    n_im = int(re.search(r"udh(\d+)", k).group(1))
                    try:
                        scores, cpos, rng = eng.wip_udh(sc, ps, n_im)
                        want = fx[f"wip_{tag}_udh{n_im}_cpos"].reshape(-1, 10)
                        wr = fx["wip_%s_udh%d_rng" % (tag, n_im)][:4].tolist()
                        ok = int(scores[0]) == int(fx[k][0]) and rng[0].tolist() == wr
                        okc = all((cpos[0][i][:4] == want[i][:4]).all() or (want[i][0] > 2**30 and cpos[0][i][0] > 2**30)
                                  for i in range(n_im + 1))
                        line.append("udh%d OK" % n_im if ok and okc else "udh%d BAD scr %d vs %d rng %s vs %s cpos %s want %s" % (
                            n_im, int(scores[0]), int(fx[k][0]), rng[0].tolist(), wr, cpos[0][:, :4].tolist(), want[:, :4].tolist()))
                    except Exception as e:
                        line.append(f"udh{n_im} EXC {e}")
            print(" | ".join(line), file=out, flush=True)
    out.close()


if __name__ == "__main__":
    try:
        main()
    except Exception:
        traceback.print_exc()
        sys.exit(1)
