#!/bin/bash
# round 4: A/B of the helper-loop variants at 32 and 64 clips, then the stage pipeline's parity tests on the variant named in $PARITY
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
(
EXTRA="--steps 1 --warmup 1 --seconds 0.25" bash scripts/gpu_ab.sh
EXTRA="--clips 64 --steps 1 --warmup 1 --seconds 0.25" bash scripts/gpu_ab.sh | sed 's/^/clips64 /'
) 2>&1 | tee gpurun_out/r04/ab_m.log
if [ -n "$PARITY" ]; then
cp mimikit_amd/libmmk_hip.so /tmp/libmmk_keep.so
cp mimikit_amd/variants/libmmk_$PARITY.so mimikit_amd/libmmk_hip.so
timeout 1200 python -m pytest tests/test_gpu_networks.py tests/test_gpu_baseline_configs.py -q -x -k "stage_pipeline or cfg4" 2>&1 | tail -5
cp /tmp/libmmk_keep.so mimikit_amd/libmmk_hip.so
fi
