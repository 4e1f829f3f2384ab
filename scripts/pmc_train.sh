#!/bin/bash
# PMC wait classes of the three training kernels (scripts/probe_train.py).  usage: bash scripts/pmc_train.sh <tag>
set -u
TAG=${1:-r02}
OUT=/tmp/w/pmct_$TAG; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$name -o $name -- python3 $R/scripts/probe_train.py > $OUT/$name.log 2>&1; echo "train $name rc=$?"; }
run wait SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32
cd $R
{ echo "# rocprofv3 --pmc passes of scripts/probe_train.py, tag $TAG";
  for p in wait mfma; do for k in mlp_bwd_dx mlp_bwd_dw "mlp_forward_kernel<false, true>"; do python3 scripts/rocpd_pmc.py $OUT/$p/${p}_results.db "$k" 2>/dev/null; done; done; } > gpurun_out/${TAG}_pmc_train.txt
cat gpurun_out/${TAG}_pmc_train.txt
