#!/bin/bash
# A/B of library variants (mimikit_amd/variants/libmmk_*.so) on a bench workload: the roofline leg's figure, two runs each, interleaved
#   WORKLOAD=stft KEY=avg_launch_us bash scripts/gpu_ab.sh
export TMPDIR=/tmp
WL=${WORKLOAD:-wavenet_cfg4}
KEY=${KEY:-us_per_step_in_kernel}
EXTRA=${EXTRA:---steps 1 --warmup 1 --seconds 0.25}
cp mimikit_amd/libmmk_hip.so /tmp/libmmk_base.so
for rep in 1 2; do
for v in mimikit_amd/variants/libmmk_*.so; do
  cp $v mimikit_amd/libmmk_hip.so
  r=$(timeout 300 python bench.py --workload $WL --no-cpu-baseline $EXTRA 2>/dev/null | grep -o "\"$KEY\": [0-9.]*")
  echo "$(basename $v) $r"
done
done
cp /tmp/libmmk_base.so mimikit_amd/libmmk_hip.so
