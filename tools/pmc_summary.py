#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters (csv output).  Usage:
    python tools/pmc_summary.py gpurun_out/pmc1/p1_counter_collection.csv [more.csv ...]"""
import csv
import collections
import sys

for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    seen = set()
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"][:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (k, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                calls[k] += 1
    print("#", path)
    for k, d in acc.items():
        print(f"{k}  calls={calls[k]}")
        for c, v in sorted(d.items()):
            print(f"    {c:28s} {v:18.0f}")
