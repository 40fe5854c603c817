#!/bin/bash
# round 4: the batched (matrix-pipe) biases of the four-behind mode: parity first, then A/B at 40 / 64 / 128 clips
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_networks.py tests/test_gpu_baseline_configs.py -q -x -k "stage_pipeline or cfg4" 2>&1 | tail -12
(
EXTRA="--clips 40 --steps 1 --warmup 1 --seconds 0.25" bash scripts/gpu_ab.sh | sed 's/^/clips40 /'
EXTRA="--clips 64 --steps 1 --warmup 1 --seconds 0.25" bash scripts/gpu_ab.sh | sed 's/^/clips64 /'
EXTRA="--clips 128 --steps 1 --warmup 1 --seconds 0.25" bash scripts/gpu_ab.sh | sed 's/^/clips128 /'
) 2>&1 | tee gpurun_out/r04/ab_o.log
