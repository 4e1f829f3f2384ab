#!/bin/bash
# PMC passes of the split-f16 backward (separate runs, one counter group each): mlp_bwd_dw_x2_kernel and mlp_bwd_dx_f16x2_kernel.
# usage (GPU box, repo root):  bash scripts/pmc_bwd_x2.sh <tag>
set -u
TAG=${1:-r06}
OUT=/tmp/w/pmcbx2_$TAG; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$name -o $name -- python3 $R/scripts/f16x2_backward_time.py 3 > $OUT/$name.log 2>&1; echo "bwd x2 $name rc=$?"; }
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16
run wait SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
{ echo "# rocprofv3 --pmc passes of scripts/f16x2_backward_time.py 3 (M = 262144 and 786432, 4 launches each per kernel), tag $TAG";
  for k in mlp_bwd_dw_x2 mlp_bwd_dx_f16x2; do echo "## $k"; for p in mfma wait lds fetch write; do python3 scripts/rocpd_pmc.py $OUT/$p/${p}_results.db $k 2>/dev/null; done; done; } > gpurun_out/${TAG}_pmc_bwd_x2.txt
cat gpurun_out/${TAG}_pmc_bwd_x2.txt
