#!/bin/bash
# round 4: A/B of the library variants on cfg 4 (32 and 64 clips), and the parity of ONE variant (VARIANT=name) on the stage-pipeline tests
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
bash scripts/gpu_ab.sh 2>&1 | tail -20
EXTRA="--steps 1 --warmup 1 --seconds 0.25 --clips 64" bash scripts/gpu_ab.sh 2>&1 | tail -20
if [ -n "$VARIANT" ]; then
  cp mimikit_amd/libmmk_hip.so /tmp/libmmk_keep.so
  cp mimikit_amd/variants/libmmk_$VARIANT.so mimikit_amd/libmmk_hip.so
  timeout 1500 python -m pytest tests -m gpu -x -q -k "${PYTEST_K:-stage_pipeline}" 2>&1 | tail -6
  cp /tmp/libmmk_keep.so mimikit_amd/libmmk_hip.so
fi
