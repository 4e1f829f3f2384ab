#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database: per-kernel calls / total / average / min / max.

    python scripts/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# source: {path}")
    print(f"{'calls':>7} {'total_ms':>11} {'avg_us':>11} {'min_us':>11} {'max_us':>11} {'pct':>6}  kernel")
    for name, calls, tot, avg, mn, mx in rows:
        print(f"{calls:7d} {tot/1e6:11.3f} {avg/1e3:11.2f} {mn/1e3:11.2f} {mx/1e3:11.2f} {100*tot/total:6.2f}  {name[:150]}")


if __name__ == "__main__":
    main(sys.argv[1])
