#!/bin/bash
# S2S parity tests, then the cfg-5 bench (twice)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 800 python -m pytest tests -q -m gpu -x -k "s2s or seq2seq or cfg5 or Seq2Seq" 2>&1 | tail -5
for i in 1 2; do
  timeout 300 python bench.py --workload s2s_cfg5 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|"us_per_generate_step": [0-9.]*' | tr '\n' ' '; echo
done
