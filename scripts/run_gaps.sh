set -u
R=$GRAFT_REPO_ROOT
mkdir -p /tmp/w && cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/w/gaps -o gaps -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-frame --no-stages --no-bf16 > /tmp/w/gaps.log 2>&1
cd $R; python3 scripts/trace_gaps.py $(find /tmp/w/gaps -name "*_results.db" | head -1)
