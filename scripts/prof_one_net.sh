#!/bin/bash
# kernel trace of scripts/layered_time.py for ONE network of the layered family.  usage (GPU box): bash scripts/prof_one_net.sh 786432 63x64
set -u
R=$GRAFT_REPO_ROOT
OUT=/tmp/w/prof_one_$2; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $R/scripts/layered_time.py $1 $2 > $OUT/trace.log 2>&1; echo "trace rc=$?"
cd $R
python3 scripts/rocpd_stats.py $(find $OUT/trace -name "*_results.db" | head -1) | head -24 | cut -c1-200
