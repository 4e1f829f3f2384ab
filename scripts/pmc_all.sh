#!/bin/bash
# All PMC passes of the round (render bench + training probe), summarised on the box into two small text files.
# usage (GPU box, repo root):  bash scripts/pmc_all.sh <tag>
set -u
TAG=${1:-r01}
OUT=/tmp/w/pmc_$TAG; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
runf() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/f_$name -o $name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --no-bf16 > $OUT/f_$name.log 2>&1; echo "fwd $name rc=$?"; }
runb() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/b_$name -o $name -- python3 $R/scripts/probe_train.py > $OUT/b_$name.log 2>&1; echo "bwd $name rc=$?"; }
runf fetch FETCH_SIZE
runf write WRITE_SIZE
runf mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
runf lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS
runb fetch FETCH_SIZE
runb write WRITE_SIZE
runb mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT
cd $R
for p in fetch write mfma lds; do python3 scripts/rocpd_pmc.py $OUT/f_$p/${p}_results.db "mlp_forward_kernel<false, false>" 2>/dev/null; done > gpurun_out/${TAG}_pmc_forward.txt
for p in fetch write mfma; do for k in mlp_bwd_dx mlp_bwd_dw mlp_forward; do python3 scripts/rocpd_pmc.py $OUT/b_$p/${p}_results.db $k 2>/dev/null; done; done > gpurun_out/${TAG}_pmc_backward.txt
wc -l gpurun_out/${TAG}_pmc_forward.txt gpurun_out/${TAG}_pmc_backward.txt
