#!/bin/bash
# PMC passes for the training kernels (probe_train.py: record-mode forward + backward at both launch shapes)
set -u
OUT=/tmp/w/pmc_bwd; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$name -o $name -- python3 $GRAFT_REPO_ROOT/scripts/probe_train.py > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT
cd $GRAFT_REPO_ROOT
for p in fetch write mfma; do for k in mlp_bwd_dx mlp_bwd_dw mlp_forward; do python scripts/rocpd_pmc.py $OUT/$p/${p}_results.db $k 2>/dev/null; done; done > gpurun_out/r01_pmc_backward.txt
