#!/bin/bash
# Round 6: kernel trace of five training steps on the coord_encode_level 12 scene pair after the thin rows moved into the dW
# list kernel (side jobs) and the 96-wide position window got its own shape; plus the step time without the profiler.
set -u
R=$GRAFT_REPO_ROOT
OUT=/tmp/w/prof_enc6; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --stats --kernel-trace -d $OUT/trace -o trace -- python3 $R/scripts/probe_encoders_train.py coord_l12 > $OUT/trace.log 2>&1; echo "trace rc=$?"
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- python3 scripts/probe_encoders_train.py coord_l12  (5 training steps, NeRF(75,27,256) behind PositionalEncoder(3,12) / (3,4); round 6: thin rows as side jobs of the dW list kernel, three-block position window)"
  python3 scripts/rocpd_stats.py $(find $OUT/trace -name "*_results.db" | head -1) | cut -c1-170 | head -12; } > gpurun_out/r06_encoders_train_trace.txt
cat gpurun_out/r06_encoders_train_trace.txt
python3 - <<'PY'
import sys, os, time, json
sys.path[:0] = [os.environ["GRAFT_REPO_ROOT"], os.path.join(os.environ["GRAFT_REPO_ROOT"], "torch-nerf_amd")]
import torch, bench
r = bench.encoder_variants_leg(torch.device("cuda", 0), 0, 10, 2)
print(json.dumps({k: {"ms": v["ms_per_step"], "train_ms": v["train"]["ms_per_step"], "train_frac": v["train"]["mlp_frac"]} for k, v in r.items() if isinstance(v, dict)}))
PY
