#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel trace) as a text table:
per kernel name: calls, total / mean / min / max duration.  Usage:
    python tools/prof_summary.py gpurun_out/prof/xxx_results.db > profiles/xxx_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
    scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    q = (f"select s.{name_col}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
         f"max(d.end - d.start), max(d.grid_size_x), max(d.workgroup_size_x) from {disp} d join {sym} s on d.kernel_id = s.id "
         f"group by s.{name_col} order by 3 desc")
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>10s} {'min_ms':>10s} {'max_ms':>10s} {'%':>6s} {'grid':>9s} {'wg':>4s}")
    for name, n, t, a, mn, mx, gx, wx in rows:
        print(f"{name[:70]:70s} {n:6d} {t / 1e6:10.3f} {a / 1e6:10.3f} {mn / 1e6:10.3f} {mx / 1e6:10.3f} {100.0 * t / tot:6.2f} {gx:9d} {wx:4d}")


if __name__ == "__main__":
    main(sys.argv[1])
