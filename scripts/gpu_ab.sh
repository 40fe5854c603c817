#!/bin/bash
# A/B of library variants (mimikit_amd/variants/libmmk_*.so) on the cfg-4 bench: in-kernel us per step, two runs each, interleaved
export TMPDIR=/tmp
cp mimikit_amd/libmmk_hip.so /tmp/libmmk_base.so
for rep in 1 2; do
for v in mimikit_amd/variants/libmmk_*.so; do
  cp $v mimikit_amd/libmmk_hip.so
  r=$(timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --seconds 0.25 2>/dev/null | grep -o '"us_per_step_in_kernel": [0-9.]*')
  echo "$(basename $v) $r"
done
done
cp /tmp/libmmk_base.so mimikit_amd/libmmk_hip.so
