#!/bin/bash
# round 4: the clips curve (us per step in the kernel) of library variants
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
cp mimikit_amd/libmmk_hip.so /tmp/libmmk_base.so
for v in mimikit_amd/variants/libmmk_*.so; do
  cp $v mimikit_amd/libmmk_hip.so
  for c in ${CLIPS:-8 16 24 28 32 36 40 48 64 96 128}; do
    r=$(timeout 300 python bench.py --clips $c --no-cpu-baseline --steps 1 --warmup 1 --seconds 0.128 2>/dev/null | grep -o "\"us_per_step_in_kernel\": [0-9.]*" | grep -o "[0-9.]*$")
    echo "$(basename $v) clips $c us_per_step $r"
  done
done 2>&1 | tee gpurun_out/r04/clips_curve_variants.log
cp /tmp/libmmk_base.so mimikit_amd/libmmk_hip.so
