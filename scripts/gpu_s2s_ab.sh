#!/bin/bash
# S2S parity tests on the built library, then the A/B of the variants
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -x -k "s2s or seq2seq or cfg5 or Seq2Seq" 2>&1 | tail -3
WORKLOAD=s2s_cfg5 KEY=us_per_generate_step EXTRA="--steps 3 --warmup 1" bash scripts/gpu_ab.sh
