#!/bin/bash
# build a variant of libmmk_hip.so whose wavenet_spipe.o is compiled with extra -D flags: scripts/build_variant.sh NAME "-DX=1 -DY=2"
set -e
cd "$(dirname "$0")/.."
mkdir -p mimikit_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $2 -c mimikit_amd/csrc/wavenet_spipe.hip -o mimikit_amd/variants/spipe_$1.o
objs=$(ls mimikit_amd/build/*.o | grep -v wavenet_spipe.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o mimikit_amd/variants/libmmk_$1.so $objs mimikit_amd/variants/spipe_$1.o
echo built $1
