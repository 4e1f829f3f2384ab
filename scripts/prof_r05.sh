#!/bin/bash
# Round-5 evidence, collected on ONE box and summarised into gpurun_out/r05_*.txt (copied into profiles/):
#   kernel trace of the default bench legs, PMC passes of the fused render kernel, PMC + trace of the HBM-bound stage
#   kernels at frame scale (640 000 rays), kernel trace of the non-default-encoder chain, the full bench line.
# usage (GPU box, repo root):  bash scripts/prof_r05.sh
set -u
TAG=r05
R=$GRAFT_REPO_ROOT
OUT=/tmp/w/prof_$TAG; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-frame --no-stages --no-traffic --no-configs --no-runner-loop"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1; echo "trace rc=$?"
pmc() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/f_$name -o $name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --no-bf16 --no-frame --no-stages --no-traffic --no-configs --no-runner-loop > $OUT/f_$name.log 2>&1; echo "fused $name rc=$?"; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS
trn() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/t_$name -o $name -- python3 $R/scripts/probe_train.py > $OUT/t_$name.log 2>&1; echo "train $name rc=$?"; }
trn wait SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
trn mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32
trn fetch FETCH_SIZE
trn write WRITE_SIZE
# the HBM-bound stages at frame scale
stg() { name=$1; shift; timeout 300 rocprofv3 "$@" --kernel-trace -d $OUT/s_$name -o $name -- python3 $R/scripts/probe_stages.py 640000 4 > $OUT/s_$name.log 2>&1; echo "stages $name rc=$?"; }
stg trace --stats
stg wait --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
stg mem --pmc SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS
stg fetch --pmc FETCH_SIZE
stg write --pmc WRITE_SIZE
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- $BENCH"; python3 scripts/rocpd_stats.py $(find $OUT/trace -name "*_results.db" | head -1) | head -30; } > gpurun_out/${TAG}_kernel_stats.txt
{ echo "# rocprofv3 --pmc passes (separate runs), bench.py --steps 3 --warmup 1, render leg only"
  echo "# kernel render_fused_kernel<1> (shipped encoders); 8 dispatches = 4 x (coarse pass 4096 x 64, fine pass 4096 x 192); FETCH_SIZE / WRITE_SIZE in KiB (raw)"
  for p in fetch write mfma lds; do python3 scripts/rocpd_pmc.py $OUT/f_$p/${p}_results.db render_fused_kernel 2>/dev/null; done; } > gpurun_out/${TAG}_pmc_render_fused.txt
{ echo "# rocprofv3 --pmc passes of scripts/probe_train.py (record forward, dX chain, dW GEMMs at 786 432 samples); FETCH_SIZE / WRITE_SIZE in KiB (raw)";
  for p in wait mfma fetch write; do for k in mlp_bwd_dx mlp_bwd_dw "mlp_forward_kernel"; do python3 scripts/rocpd_pmc.py $OUT/t_$p/${p}_results.db "$k" 2>/dev/null; done; done; } > gpurun_out/${TAG}_pmc_train.txt
{ echo "# scripts/probe_stages.py 640000 4: the HBM-bound stage kernels at frame scale (640 000 rays x 64 / 64+128 samples)"
  echo "# -- kernel trace (rocprofv3 --kernel-trace --stats)"; python3 scripts/rocpd_stats.py $(find $OUT/s_trace -name "*_results.db" | head -1) | head -14
  echo "# -- PMC passes (separate runs); FETCH_SIZE / WRITE_SIZE in KiB (raw)"
  for p in wait mem fetch write; do for k in stratified_kernel hierarchical_kernel composite_fwd composite_bwd; do python3 scripts/rocpd_pmc.py $OUT/s_$p/${p}_results.db "$k" 2>/dev/null; done; done; } > gpurun_out/${TAG}_pmc_stages.txt
bash scripts/pmc_bf16.sh $TAG > /dev/null 2>&1
python3 bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
wc -l gpurun_out/${TAG}_*.txt; tail -c 400 gpurun_out/${TAG}_bench_n1.json
