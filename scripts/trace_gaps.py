"""Where one training step's time goes: per-kernel totals and the idle gaps between kernels, from a rocprofv3
--kernel-trace rocpd database.   python scripts/trace_gaps.py <results.db> [first_kernel_substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
# steps = runs between successive nerf_adam kernels
ends = [i for i, r in enumerate(rows) if "adam" in r[0]]
if len(ends) < 4:
    print("not enough steps"); sys.exit(0)
lo, hi = ends[-3] + 1, ends[-2] + 1          # one steady-state step
step = rows[lo:hi]
t0, t1 = step[0][1], step[-1][2]
busy = sum(e - s for _, s, e in step)
print(f"step: {len(step)} kernels, wall {(t1 - t0) / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms")
agg = {}
for n, s, e in step:
    k = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {c:3d} x {t / 1e3:9.1f} us  {k}")
big = [(n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40], (e - s_) / 1e3) for n, s_, e in step if e - s_ > 5e5]
print("launches > 0.5 ms, in order:", ", ".join(f"{n} {t:.0f}" for n, t in big))
gaps = sorted(((step[i + 1][1] - step[i][2], step[i][0].replace("(anonymous namespace)::", "")[:40], step[i + 1][0].replace("(anonymous namespace)::", "")[:40]) for i in range(len(step) - 1)), reverse=True)
print("largest gaps:")
for g, a, b in gaps[:8]:
    print(f"  {g / 1e3:7.1f} us  after {a}  before {b}")
