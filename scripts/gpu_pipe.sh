#!/bin/bash
# pipelined WaveNet kernel: parity subset, cfg4 at its real size, bench + stamps
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=300 -x -k "pipelined or modes_agree" > gpurun_out/pytest_pipe.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_pipe.log
tail -5 gpurun_out/pytest_pipe.log
timeout 600 python -m pytest tests/test_gpu_baseline_configs.py -m gpu -q --timeout=500 -x -k "cfg4" > gpurun_out/pytest_cfg4.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_cfg4.log
tail -4 gpurun_out/pytest_cfg4.log
MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 MMK_WN_STAMP_STAGE=${STAGE:-7} MMK_WN_STAMP_OWNER=${OWNER:-0} timeout 300 python bench.py --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps" | tail -1 | cut -c150-1300
timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --seconds 0.25 > gpurun_out/bench_wn.json 2> gpurun_out/bench_wn.err; echo "bench exit $?"
grep -o '"value".\{0,30\}\|"us_per_ar_step".\{0,10\}\|"us_per_step_in_kernel".\{0,10\}' gpurun_out/bench_wn.json
