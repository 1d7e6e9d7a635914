"""the live_h_* fixtures on the GPU, with the oracle's per-request results beside the device's"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import spdg
from tests.conftest import golden_files
from tests.test_oracle_seeded_h import seeded_inputs_h
from spaln_amd import engine
from oracle import seeded
eng = engine.Engine(0)
for path in golden_files("live_h_"):
    fx = spdg.load(path)
    sc, sp, p, hsps, n, lowest, wl = seeded_inputs_h(fx, 0)
    sc.scalar_engines = 1
    for pipe in (None, "0", "1"):
        if pipe is None: os.environ.pop("SPDP_A0_PIPE", None)
        else: os.environ["SPDP_A0_PIPE"] = pipe
        (scr, skl), = eng.align_h_seeded(sc, sp, p._owner, [hsps if n else None], [lowest], [wl])
        flat = [int(x) for x in skl.ravel()]
        want = fx["seed_skl_A0"].tolist()
        print(path.split("/")[-1], "pipe", pipe, "score", scr, int(fx["seed_scr_A0"][0]), "equal" if flat == want else "DIFFERENT")
        if flat != want:
            print("  got ", flat); print("  want", want)
