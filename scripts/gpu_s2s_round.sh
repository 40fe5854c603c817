#!/bin/bash
# S2S parity tests, phase stamps of the resident kernel, A/B of the variants
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -x -k "s2s or seq2seq or cfg5 or Seq2Seq" 2>&1 | tail -3
bash scripts/gpu_s2s_stamps.sh ${1:-0} | cut -c1-230
WORKLOAD=s2s_cfg5 KEY=us_per_generate_step EXTRA="--steps 3 --warmup 1" bash scripts/gpu_ab.sh
