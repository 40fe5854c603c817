#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --seconds 0.25 > gpurun_out/bench_wn.json 2> gpurun_out/bench_wn.err; echo "bench exit $?"
grep -o '"value".\{0,30\}\|"us_per_ar_step".\{0,10\}\|"us_per_step_in_kernel".\{0,10\}' gpurun_out/bench_wn.json
MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 timeout 300 python bench.py --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps" | tail -1 | cut -c1-1200
