#!/bin/bash
# round 4: parity of the stage pipeline, the clips curve, A/B of the variants at 32 and 64 clips
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "${PYTEST_K:-stage_pipeline or cfg4}" 2>&1 | tail -5
for c in ${CLIPS:-8 24 28 32 36 40 48 64 128}; do
  r=$(timeout 300 python bench.py --clips $c --steps 1 --warmup 1 --seconds 0.25 --no-cpu-baseline 2>/dev/null | grep -o '"us_per_step_in_kernel": [0-9.]*')
  echo "clips $c $r"
done | tee gpurun_out/r04/clips_curve.log
bash scripts/gpu_ab.sh 2>&1 | tail -20
EXTRA="--steps 1 --warmup 1 --seconds 0.25 --clips 64" bash scripts/gpu_ab.sh 2>&1 | tail -20
