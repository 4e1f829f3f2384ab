#!/bin/bash
# Round-6 evidence that the stored kernel trace reproduces the stored bench line (verdict r05, "next" #3): ONE box,
# the same bench command three times -- plain, under rocprofv3 --kernel-trace --stats, plain again -- so that the
# profiler's inflation and the box's own drift are both on record next to the per-kernel averages.
# usage (GPU box, repo root):  bash scripts/prof_r06_samebox.sh
set -u
TAG=r06
R=$GRAFT_REPO_ROOT
OUT=/tmp/w/same_$TAG; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 40 --warmup 5 --no-cpu-baseline --no-train --no-bf16 --no-f16x2 --no-frame --no-stages --no-traffic --no-configs --no-runner-loop"
python3 $R/bench.py $ARGS > $OUT/plain_a.json 2> $OUT/plain_a.err; echo "plain a rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $R/bench.py $ARGS > $OUT/profiled.json 2> $OUT/profiled.err; echo "trace rc=$?"
python3 $R/bench.py $ARGS > $OUT/plain_b.json 2> $OUT/plain_b.err; echo "plain b rc=$?"
cd $R
{ echo "# one box, three runs of: python3 bench.py $ARGS"
  echo "# (1) plain  (2) rocprofv3 --kernel-trace --stats -- python3 bench.py ...  (3) plain again"
  python3 - $OUT <<'EOF'
import json, sys, os, subprocess, glob
out = sys.argv[1]
lines = {}
for tag in ("plain_a", "profiled", "plain_b"):
    txt = [l for l in open(os.path.join(out, tag + ".json")) if l.startswith("{")]
    lines[tag] = json.loads(txt[-1]) if txt else None
for tag, l in lines.items():
    if l is None:
        print(f"{tag}: no line"); continue
    r = l["roofline"]
    print(f"{tag:9s} ms_per_step {l['ms_per_step']:.4f}  rays/s {l['value']:.0f}  roofline.frac {r['frac']:.4f}  "
          f"ms_per_launch(events) {r['ms_per_launch']}")
db = glob.glob(os.path.join(out, "trace", "**", "*_results.db"), recursive=True)
import sqlite3
con = sqlite3.connect(db[0]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels where {name_col} like '%render_fused_kernel%' order by start").fetchall()
# the timed region = the last 2 x steps launches (warm-up launches come first)
steps = lines["profiled"]["steps"]
timed = rows[-2 * steps:]
avg_all = sum(e - s for _, s, e in rows) / len(rows) / 1e3
avg_timed = sum(e - s for _, s, e in timed) / len(timed) / 1e3
flop = 524288 * 1186816.0
print(f"trace: render_fused_kernel calls {len(rows)}  avg_us(all) {avg_all:.2f}  avg_us(timed {len(timed)}) {avg_timed:.2f}")
print(f"trace: 2 x avg(timed) = {2 * avg_timed / 1e3:.4f} ms  vs ms_per_step of the SAME run {lines['profiled']['ms_per_step']:.4f}"
      f"  and of the plain runs {lines['plain_a']['ms_per_step']:.4f} / {lines['plain_b']['ms_per_step']:.4f}")
print(f"trace: frac recomputed = {flop / (avg_timed * 1e-6) / 157.3e12:.4f}  (line under the profiler {lines['profiled']['roofline']['frac']:.4f}, "
      f"plain {lines['plain_a']['roofline']['frac']:.4f} / {lines['plain_b']['roofline']['frac']:.4f})")
EOF
  echo "# -- per-kernel summary of run (2)"
  python3 scripts/rocpd_stats.py $(find $OUT/trace -name "*_results.db" | head -1) | head -12
  echo "# -- the three lines"
  for t in plain_a profiled plain_b; do echo "## $t"; grep '^{' $OUT/$t.json | tail -1; done
} > gpurun_out/${TAG}_samebox.txt 2>&1
cat gpurun_out/${TAG}_samebox.txt | head -12
