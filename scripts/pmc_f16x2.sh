#!/bin/bash
# PMC passes of the f16x2 fused MLP kernel (separate runs, one counter group each), summarised into one text file.
# usage (GPU box, repo root):  bash scripts/pmc_f16x2.sh <tag>
set -u
TAG=${1:-r02}
OUT=/tmp/w/pmcx2_$TAG; mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$name -o $name -- python3 $R/scripts/probe_f16x2.py > $OUT/$name.log 2>&1; echo "f16x2 $name rc=$?"; }
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16
run wait SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $R/scripts/probe_f16x2.py > $OUT/trace.log 2>&1
cd $R
{ echo "# rocprofv3 --pmc passes of scripts/probe_f16x2.py (6 launches, M = 786432), tag $TAG";
  for p in mfma wait lds fetch write; do python3 scripts/rocpd_pmc.py $OUT/$p/${p}_results.db mlp_forward_f16x2 2>/dev/null; done;
  echo "# kernel trace (--stats):"; python3 scripts/rocpd_stats.py $OUT/trace/trace_results.db 2>/dev/null | head -12; } > gpurun_out/${TAG}_pmc_f16x2.txt
cat gpurun_out/${TAG}_pmc_f16x2.txt
