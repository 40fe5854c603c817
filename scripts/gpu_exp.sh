#!/bin/bash
# timing experiments on the persistent WaveNet kernel (diagnostics only)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 timeout 300 python bench.py --workload wavenet_cfg4 --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps|us_per_ar_step" | tail -2 | sed -e 's/.*"us_per_step_in_kernel"/us_per_step_in_kernel/' | cut -c1-400; }
run MMK_WN_X=0
run MMK_WN_EXPERIMENT_SAME_WEIGHTS=1
