#!/bin/bash
# Round-4 evidence for the layered family (csrc/mlp_layered.hip): kernel trace + PMC passes (MFMA busy, wait classes,
# FETCH / WRITE) of scripts/layered_time.py on NeRF(63,27,128) (narrow register-resident kernels) and NeRF(63,27,512)
# (plane-parked kernel), summarised into gpurun_out/r04_*.txt.  usage (GPU box, repo root): bash scripts/prof_layered.sh
set -u
TAG=r04
R=$GRAFT_REPO_ROOT
OUT=/tmp/w/prof_layered; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/scripts/layered_time.py 786432 75x256,128,63x64,512"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1; echo "trace rc=$?"
pmc() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
pmc wait SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- $CMD"; python3 scripts/rocpd_stats.py $(find $OUT/trace -name "*_results.db" | head -1) | head -24; } > gpurun_out/${TAG}_layered_kernel_stats.txt
{ echo "# rocprofv3 --pmc passes (separate runs) of: $CMD"
  echo "# NeRF(63,27,128) / NeRF(63,27,64): reg_forward_kernel<2,4,..> / <2,2,..>, narrow_dx_kernel<4,..> / <2,..>; NeRF(75,27,256): reg_forward_kernel<1,8,3,..> / layered_kernel<true>; NeRF(63,27,512): layered_kernel<false> (forward) / <true> (reverse chain);"
  echo "# dW of both: mlp_bwd_dw_list_kernel.  FETCH_SIZE / WRITE_SIZE in KiB (raw; FETCH_SIZE tallies 16-B/lane streams at half their bytes)"
  for p in mfma wait fetch write; do for k in reg_forward narrow_dx layered_kernel mlp_bwd_dw_list layered_thin; do python3 scripts/rocpd_pmc.py $OUT/$p/${p}_results.db "$k" 2>/dev/null; done; done; } > gpurun_out/${TAG}_pmc_layered.txt
python3 scripts/layered_time.py 262144 > gpurun_out/${TAG}_layered_family.txt 2>&1
python3 scripts/layered_time.py 786432 >> gpurun_out/${TAG}_layered_family.txt 2>&1
cat gpurun_out/${TAG}_layered_family.txt | grep -v amdgpu; head -20 gpurun_out/${TAG}_layered_kernel_stats.txt
