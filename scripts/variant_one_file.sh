#!/bin/bash
# A/B build of ONE kernel file: recompile it with extra defines and link against the other objects of the regular build.
# usage: bash scripts/variant_one_file.sh <name> <file.hip> "<-DX_FOO ...>"   ->  torch-nerf_amd/lib/variants/<name>.so
# pick it at run time with NERF_AMD_LIB=torch-nerf_amd/lib/variants/<name>.so
set -e
NAME=$1; FILE=$2; DEFS=$3
cd "$(dirname "$0")/../torch-nerf_amd/csrc"
mkdir -p build_$NAME ../lib/variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $FLAGS $DEFS -c $FILE -o build_$NAME/$FILE.o
OBJS=$(ls build/*.o | grep -v "build/$FILE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build_$NAME/$FILE.o -o ../lib/variants/$NAME.so
echo "built ../lib/variants/$NAME.so"
