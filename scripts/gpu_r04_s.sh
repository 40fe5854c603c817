#!/bin/bash
# round 4: the Seq2Seq output projection's K split (tuning MMK_GEMM_KSPLIT): which one the cfg-5 generate step wants
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
for ks in default 1 2 4 8; do
python -c "
import sys, runpy
import mimikit_amd as mmk
if '$ks' != 'default': mmk.native.PLAN_TUNING['MMK_GEMM_KSPLIT'] = '$ks'
sys.argv = ['bench.py', '--workload', 's2s_cfg5', '--no-cpu-baseline', '--steps', '6', '--warmup', '2']
runpy.run_path('bench.py', run_name='__main__')
" 2>/dev/null | grep -o '"value": [0-9.]*\|"us_per_generate_step": [0-9.]*' | tr '\n' ' ' | sed "s/^/ksplit $ks: /"; echo
done 2>&1 | tee gpurun_out/r04/s2s_ksplit.log
