#!/bin/bash
# A/B of the inline-asm sites against compiler-visible forms, ONE box,
# alternating: tree build vs -DX_DMA_BUILTIN (__builtin_amdgcn_global_load_lds instead of the asm LDS-DMA piece) vs
# -DX_STORE_PLAIN (plane stores as C++ stores instead of asm global_store_dwordx4 with a scalar base).
#   make -C torch-nerf_amd/csrc EXTRA=-DX_DMA_BUILTIN BUILD=build_dmab OUT=../lib/variants/dmab.so   (likewise stp)
# usage (GPU box, repo root): bash scripts/ab_asm.sh [rounds]
N=${1:-2}
one() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-frame --no-stages --no-traffic --no-configs --no-runner-loop --no-bf16 2>/dev/null | python -c "
import sys, json
d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; t=d['train']; r=t['roofline']
print('$1', 'render rays/s', round(d['value']), 'frac', d['roofline']['frac'], '| train ms', round(t['ms_per_step'],3), 'frac', r['frac'], 'fwd', r['forward_record']['ms_per_step'], 'bwd', r['backward']['ms_per_step'])"; }
for i in $(seq $N); do
  one tree
  NERF_AMD_LIB=$PWD/torch-nerf_amd/lib/variants/dmab.so one dma_builtin
  NERF_AMD_LIB=$PWD/torch-nerf_amd/lib/variants/stp.so one store_plain
done
# the variants must still be right: the training-parity and determinism tests on each
for v in dmab stp; do
  NERF_AMD_LIB=$PWD/torch-nerf_amd/lib/variants/$v.so python -m pytest tests/test_gpu_backward.py tests/test_gpu_fused.py tests/test_gpu_determinism.py -q -m gpu -x 2>&1 | tail -2 | sed "s/^/$v: /"
done
