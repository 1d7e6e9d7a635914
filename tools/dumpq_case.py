#!/usr/bin/env python3
"""Dev tool (build container only): the fixture of ONE query of a tools/dropin_demo.py run, written by the reference's own
CLI from inside its alignH_ng call (oracle/_ref/spaln_dumpq, oracle/ref_build/dumpq.cc).

    python tools/dumpq_case.py --queries 10000 --genes 200 --protein --name q7555 --out /tmp/q7555.spdg [--mode Q7]

The data set is the one dropin_demo.py makes for the same --queries / --genes (same seeds), so a record that differed
there can be taken apart here: the fixture holds the pair as blkaln handed it over (window, Exinon arrays, HSPs), every
Wilip reply and the reference's score + SKL for -A0 and -A2 -- what tests/test_oracle_seeded.py and the GPU tests read."""
import argparse
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import dropin_demo  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=10000)
    ap.add_argument("--genes", type=int, default=200)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--protein", action="store_true")
    ap.add_argument("--name", required=True)
    ap.add_argument("--call", type=int, default=0)
    ap.add_argument("--mode", default="Q7")
    ap.add_argument("--out", required=True)
    ap.add_argument("--spacer", type=int, default=0)
    ap.add_argument("--frag", type=int, default=0)
    ap.add_argument("--ori", type=int, default=1, choices=[1, 3], help="3: the data set of tools/e2e_q7.py --ori 3 (every other query reverse-complemented, no -S1)")
    ap.add_argument("--keep", default="", help="directory to make the data set in (kept); default: a temporary one")
    args = ap.parse_args()
    td = args.keep or tempfile.mkdtemp(prefix="spdp_dumpq_")
    os.makedirs(td, exist_ok=True)
    if not os.path.exists(os.path.join(td, "gnm.bkp" if args.protein else "gnm.bkn")):
        _, env = dropin_demo.make_dataset(td, args)
        if args.ori == 3:                                        # (as tools/e2e_q7.py does)
            lines = open(os.path.join(td, "q.fa")).read().split("\n")
            comp = str.maketrans("ACGTacgt", "TGCAtgca")
            for i in range(2, len(lines) - 1, 4):
                lines[i + 1] = lines[i + 1].translate(comp)[::-1]
            open(os.path.join(td, "q.fa"), "w").write("\n".join(lines))
    else:
        env = dict(os.environ, ALN_TAB=os.path.join(dropin_demo.REF, "table"), ALN_DBS=td)
    # the one query on its own (the block search of a query does not depend on the others)
    lines = open(os.path.join(td, "q.fa")).read().split(">")
    one = [x for x in lines if x.startswith(args.name + "\n")]
    if not one:
        raise SystemExit(f"{args.name} is not in q.fa")
    with open(os.path.join(td, "one.fa"), "w") as f:
        f.write(">" + one[0])
    env = dict(env, DUMPQ_NAME=args.name, DUMPQ_OUT=os.path.abspath(args.out), DUMPQ_CALL=str(args.call))
    r = subprocess.run([os.path.join(dropin_demo.REF, "spaln_dumpq"), "-" + args.mode] + ([] if args.ori == 3 or args.protein else ["-S1"]) + ["-O4", "-t1", "-dgnm", "one.fa"], cwd=td, env=env,
                       capture_output=True, text=True)
    print(r.stderr[-600:])
    print(r.stdout[-1500:])
    print("data set in", td)


if __name__ == "__main__":
    main()
