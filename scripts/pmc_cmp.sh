set -u
OUT=/tmp/w/pmc_cmp; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM --kernel-trace -d $OUT/a -o a -- python3 $GRAFT_REPO_ROOT/scripts/probe_mlp.py > $OUT/a.log 2>&1; echo rc=$?
timeout 300 rocprofv3 --pmc SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --kernel-trace -d $OUT/b -o b -- python3 $GRAFT_REPO_ROOT/scripts/probe_mlp.py > $OUT/b.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT; for p in a b; do python scripts/rocpd_pmc.py $OUT/$p/${p}_results.db mlp_forward 2>/dev/null; done
