#!/bin/bash
# build a variant of libmmk_hip.so with ONE source compiled with extra -D flags: scripts/build_variant.sh NAME "-DX=1 -DY=2" [source.hip]
set -e
cd "$(dirname "$0")/.."
src=${3:-wavenet_spipe.hip}
mkdir -p mimikit_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -c mimikit_amd/csrc/$src -o mimikit_amd/variants/${src%.hip}_$1.o
objs=$(ls mimikit_amd/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mimikit_amd/variants/libmmk_$1.so $objs mimikit_amd/variants/${src%.hip}_$1.o
echo built $1
