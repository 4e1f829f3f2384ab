#!/bin/bash
# PMC passes + kernel trace of the training step on the coord_encode_level 12 scene pair (NeRF(75,27,256), layered family:
# reg_forward_kernel<1,8,3,true,1>, reg_dx_kernel, mlp_bwd_dw_list_kernel, layered_thin_kernel, encode_to_plane_kernel).
# usage (GPU box, repo root): bash scripts/prof_encoders.sh
set -u
R=$GRAFT_REPO_ROOT
OUT=/tmp/w/prof_enc; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 "$@" --kernel-trace -d $OUT/$name -o $name -- python3 $R/scripts/probe_encoders_train.py coord_l12 > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run trace --stats
run mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32
run wait --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
cd $R
{ echo "# scripts/prof_encoders.sh: 5 training steps, coord_encode_level 12 (NeRF(75,27,256) behind PositionalEncoder(3,12) / (3,4)), 4096 rays x (64 + 192) samples"
  echo "# -- kernel trace"; python3 scripts/rocpd_stats.py $(find $OUT/trace -name "*_results.db" | head -1) | cut -c1-160 | head -10
  echo "# -- PMC passes (separate runs); FETCH_SIZE / WRITE_SIZE in KiB (raw)"
  for p in mfma wait fetch write; do for k in reg_forward_kernel reg_dx_kernel mlp_bwd_dw_list_kernel layered_thin_kernel encode_to_plane_kernel; do python3 scripts/rocpd_pmc.py $OUT/$p/${p}_results.db "$k" 2>/dev/null; done; done; } > gpurun_out/r05_pmc_encoders_train.txt
wc -l gpurun_out/r05_pmc_encoders_train.txt
