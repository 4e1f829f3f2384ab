#!/bin/bash
# Round-6 evidence, ONE box: (1) the same-box triple of the headline (plain / under rocprofv3 / plain), (2) kernel trace of
# the bench's render + f16x2 + bf16 + train legs, (3) PMC passes of the fused render kernel and of the split-f16 kernel,
# (4) the full bench line.  Summaries land in gpurun_out/r06_*.txt|json (copied into profiles/).
# usage (GPU box, repo root):  bash scripts/prof_r06.sh
set -u
TAG=r06
R=$GRAFT_REPO_ROOT
OUT=/tmp/w/prof_$TAG; mkdir -p $OUT $R/gpurun_out
bash $R/scripts/prof_r06_samebox.sh
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-frame --no-stages --no-traffic --no-configs --no-runner-loop"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1; echo "trace rc=$?"
pmc() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/f_$name -o $name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --no-bf16 --no-f16x2 --no-frame --no-stages --no-traffic --no-configs --no-runner-loop > $OUT/f_$name.log 2>&1; echo "fused $name rc=$?"; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- $BENCH"; python3 scripts/rocpd_stats.py $(find $OUT/trace -name "*_results.db" | head -1) | head -30; } > gpurun_out/${TAG}_kernel_stats.txt
{ echo "# rocprofv3 --pmc passes (separate runs), bench.py --steps 3 --warmup 1, render leg only"
  echo "# kernel render_fused_kernel<1> (shipped encoders); 8 dispatches = 4 x (coarse pass 4096 x 64, fine pass 4096 x 192); FETCH_SIZE / WRITE_SIZE in KiB (raw)"
  for p in fetch write mfma; do python3 scripts/rocpd_pmc.py $OUT/f_$p/${p}_results.db render_fused_kernel 2>/dev/null; done; } > gpurun_out/${TAG}_pmc_render_fused.txt
bash scripts/pmc_f16x2.sh $TAG > /dev/null 2>&1
python3 bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
wc -l gpurun_out/${TAG}_*.txt; tail -c 600 gpurun_out/${TAG}_bench_n1.json; tail -5 gpurun_out/${TAG}_bench_n1.err
bash scripts/prof_r06_encoders.sh > /dev/null 2>&1; head -8 gpurun_out/r06_encoders_train_trace.txt | cut -c1-150
