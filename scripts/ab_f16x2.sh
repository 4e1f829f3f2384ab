#!/bin/bash
# A/B of mlp_forward_f16x2 variants (scripts/variant_one_file.sh) on one box: parity subset + kernel time each.
R=$GRAFT_REPO_ROOT; cd $R
for lib in default $(ls torch-nerf_amd/lib/variants/f2_*.so 2>/dev/null); do
  if [ "$lib" = default ]; then unset NERF_AMD_LIB; else export NERF_AMD_LIB=$R/$lib; fi
  echo "=== $lib"
  python -m pytest tests/test_gpu_f16x2.py -x -q -k "golden_f5 or tile_shape or golden_f7" 2>&1 | tail -1
  python scripts/f16x2_time.py 20 2>&1 | grep f16x2
done
