#!/usr/bin/env python3
"""development aid: compare two SPDP_SEED_DUMP files (see tools/dbg/seed_dump_run.sh) block by block, query by query"""
import struct
import sys

import numpy as np


def load(path):
    raw = open(path, "rb").read()
    recs, cur, at = [], None, 0
    while at < len(raw):
        name = raw[at:at + 32].split(b"\0")[0].decode()
        n, = struct.unpack_from("<Q", raw, at + 32)
        data = raw[at + 40:at + 40 + n]
        at += 40 + n
        if name == "scoring":
            cur = {}
            recs.append(cur)
        cur[name] = data
    return recs


def key(r):
    return (len(r["a"]), r["a"][:64])


def main():
    A, B = load(sys.argv[1]), load(sys.argv[2])
    print(len(A), "records in", sys.argv[1], ";", len(B), "in", sys.argv[2])
    for ra in A:
        for rb in B:
            if ra["a"] != rb["a"]:
                continue
            pa, pb = np.frombuffer(ra["problem"], dtype=np.int32), np.frombuffer(rb["problem"], dtype=np.int32)
            print("query of", len(ra["a"]), "nt; problem ints", pa.tolist(), "|", pb.tolist())
            for k in sorted(set(ra) | set(rb)):
                xa, xb = ra.get(k, b""), rb.get(k, b"")
                if xa == xb:
                    continue
                dt = np.int16 if k in ("sig5", "sig3", "intpen") else (np.int32 if k in ("scoring", "seed", "problem", "hsps", "lowest", "wilip") else np.uint8)
                va, vb = np.frombuffer(xa, dtype=dt), np.frombuffer(xb, dtype=dt)
                if len(va) != len(vb):
                    print("  ", k, "lengths differ", len(va), len(vb))
                    m = min(len(va), len(vb))
                    va, vb = va[:m], vb[:m]
                d = np.nonzero(va != vb)[0]
                print("  ", k, len(d), "elements differ; first", [(int(i), int(va[i]), int(vb[i])) for i in d[:12]])


if __name__ == "__main__":
    main()
