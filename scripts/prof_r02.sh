#!/bin/bash
# Round-2 evidence, collected on ONE box and summarised into gpurun_out/<tag>_*.txt (copy into profiles/):
#   kernel trace of the default bench, PMC passes of the fused render kernel and of the bf16 MLP kernel,
#   the LDS-DMA stream microbenchmark, the zero-weights A/B (power-bound check), the training kernel trace.
# usage (GPU box, repo root):  bash scripts/prof_r02.sh <tag>
set -u
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=/tmp/w/prof_$TAG; mkdir -p $OUT $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-frame --no-stages"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1; echo "trace rc=$?"
pmc() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/f_$name -o $name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --no-bf16 --no-frame --no-stages > $OUT/f_$name.log 2>&1; echo "fused $name rc=$?"; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS
cd $R
{ echo "# rocprofv3 --kernel-trace --stats -- $BENCH"; python3 scripts/rocpd_stats.py $(find $OUT/trace -name "*_results.db" | head -1) | head -30; } > gpurun_out/${TAG}_kernel_stats.txt
{ echo "# rocprofv3 --pmc passes (separate runs), bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --no-bf16 --no-frame --no-stages"
  echo "# kernel render_fused_kernel; 8 dispatches = 4 x (coarse pass 4096 x 64, fine pass 4096 x 192); FETCH_SIZE / WRITE_SIZE in KiB (raw)"
  for p in fetch write mfma lds; do python3 scripts/rocpd_pmc.py $OUT/f_$p/${p}_results.db render_fused_kernel 2>/dev/null; done; } > gpurun_out/${TAG}_pmc_render_fused.txt
bash scripts/pmc_bf16.sh $TAG > /dev/null 2>&1
./scripts/ldsdma_stream.bin > gpurun_out/${TAG}_ldsdma_stream.txt 2>&1
{ echo "# scripts/ab_bf16.py: bf16 fused MLP kernel, 786 432 samples, synthetic weights vs ALL-ZERO weights (same instruction stream)";
  python3 scripts/ab_bf16.py --rounds 2; echo "# --zero"; python3 scripts/ab_bf16.py --zero --rounds 2;
  echo "# fp32 fused MLP kernel, same comparison"; python3 scripts/ab_bf16.py --fp32 --rounds 1; echo "# --zero"; python3 scripts/ab_bf16.py --fp32 --zero --rounds 1; } > gpurun_out/${TAG}_power_bound_ab.txt 2>&1
bash scripts/prof_train.sh $TAG > /dev/null 2>&1
python3 bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
wc -l gpurun_out/${TAG}_*.txt; tail -c 600 gpurun_out/${TAG}_bench_n1.json
