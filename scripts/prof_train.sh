#!/bin/bash
# Kernel trace of the device-resident training step (scripts/train_procedural.py), summarised on the box.
# usage (GPU box, repo root):  bash scripts/prof_train.sh <tag>
set -u
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p /tmp/w && cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/w/prof -o ${TAG}_train -- python3 $GRAFT_REPO_ROOT/scripts/train_procedural.py --steps 40 --size 96 --views 8 --eval-every 1000 --crop-steps 0 --out /tmp/w/tp > $OUT/${TAG}_train_prof.log 2>&1
echo "rc=$?"
DB=$(find /tmp/w/prof -name "*_results.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 scripts/train_procedural.py --steps 40 --size 96 --views 8 --eval-every 1000 --crop-steps 0"
  echo "# 40 training steps (4096 rays x (64 + 192) samples, fwd+bwd+fused Adam) + 2 held-out evaluations of 3 frames"
  python3 $GRAFT_REPO_ROOT/scripts/rocpd_stats.py $DB | head -40; } > $OUT/${TAG}_train_kernel_stats.txt
tail -3 $OUT/${TAG}_train_prof.log
