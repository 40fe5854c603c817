#!/bin/bash
# round 4: the conditioning projection as a tiled GEMM over compact, padded rows: parity of the WaveNet tests, then the bench line with and without
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1800 python -m pytest tests/test_gpu_networks.py tests/test_gpu_baseline_configs.py -q -x -k "stage_pipeline or cfg4 or wavenet or freqnet" 2>&1 | tail -5
echo "== with the GEMM"
python bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r04/bench_q.json | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"us_per_ar_step": [0-9.]*' | tr '\n' ' '; echo
echo "== MMK_WN_COND_GEMM=0"
python -c "
import sys, runpy
import mimikit_amd as mmk
mmk.native.PLAN_TUNING['MMK_WN_COND_GEMM'] = '0'
sys.argv = ['bench.py', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')
" 2>/dev/null | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"us_per_ar_step": [0-9.]*' | tr '\n' ' '; echo
