#!/usr/bin/env python3
"""Per-kernel PMC counter means from a rocprofv3 rocpd database (--pmc run).

    python scripts/rocpd_pmc.py gpurun_out/pmc_r01/fetch/fetch_results.db [kernel-substring]
"""
import sqlite3
import sys
from collections import defaultdict


def main(path, needle=""):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    print("# columns:", cols, file=sys.stderr)
    rows = cur.execute("select * from counters_collection").fetchall()
    ci = {c: i for i, c in enumerate(cols)}
    kname = next(c for c in cols if "kernel_name" in c or c == "name")
    cname = next(c for c in cols if "counter_name" in c)
    vname = next(c for c in cols if c in ("value", "counter_value"))
    did = next(c for c in cols if "dispatch_id" in c)
    acc = defaultdict(lambda: defaultdict(dict))
    for r in rows:
        k = r[ci[kname]]
        if needle and needle not in k:
            continue
        d = acc[k][r[ci[cname]]]
        d[r[ci[did]]] = d.get(r[ci[did]], 0.0) + float(r[ci[vname]])
    print(f"# source: {path}")
    for k, counters in acc.items():
        print(k[:140])
        for c, per_dispatch in sorted(counters.items()):
            vals = list(per_dispatch.values())
            print(f"    {c:34s} dispatches={len(vals):4d} mean={sum(vals)/len(vals):.6g} min={min(vals):.6g} max={max(vals):.6g}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
