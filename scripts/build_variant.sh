#!/bin/bash
# build a variant of libmmk_hip.so with ONE source compiled with extra -D flags: scripts/build_variant.sh NAME "-DX=1 -DY=2" [source.hip] [diag]
# (diag: on top of the diagnostic build's objects - python -m mimikit_amd.build --diag - with -DMMK_DIAG: a variant with phase stamps)
# The variant is loaded BY PATH: MMK_DIAG_LIB=$PWD/mimikit_amd/variants/libmmk_NAME.so (scripts/gpu_evidence.sh ab / stamps) - the product library is never replaced
set -e
cd "$(dirname "$0")/.."
src=${3:-wavenet_spipe.hip}
objdir=mimikit_amd/build; extra=""
if [ "$4" = diag ]; then objdir=mimikit_amd/build_diag; extra="-DMMK_DIAG"; fi
mkdir -p mimikit_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra $2 -c mimikit_amd/csrc/$src -o mimikit_amd/variants/${src%.hip}_$1.o
objs=$(ls $objdir/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mimikit_amd/variants/libmmk_$1.so $objs mimikit_amd/variants/${src%.hip}_$1.o
echo built $1
