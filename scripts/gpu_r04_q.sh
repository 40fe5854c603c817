#!/bin/bash
# round 4: parity of the WaveNet tests after a change outside the step kernels, then the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1800 python -m pytest tests/test_gpu_networks.py tests/test_gpu_baseline_configs.py -q -x -k "stage_pipeline or cfg4 or cfg2 or wavenet or freqnet or prefill" 2>&1 | tail -5
python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r04/bench_q.json | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"us_per_ar_step": [0-9.]*' | tr '\n' ' '; echo
