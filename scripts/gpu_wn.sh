#!/bin/bash
# WaveNet parity subset + cfg4 bench (+ stamps)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_networks.py tests/test_gpu_baseline_configs.py -m gpu -q --timeout=300 -x -k "wavenet or cfg4 or cfg2 or repack" > gpurun_out/pytest_wn.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_wn.log
tail -25 gpurun_out/pytest_wn.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_wn.json 2> gpurun_out/bench_wn.err; echo "bench exit $?"
cut -c1-400 gpurun_out/bench_wn.json; grep -o '"roofline".*' gpurun_out/bench_wn.json | cut -c1-600
MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 timeout 300 python bench.py --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps" | cut -c1-1200
timeout 300 python bench.py --workload wavenet_cfg2 --no-cpu-baseline 2>/dev/null | grep -o '"value".\{0,30\}\|"us_per_ar_step".\{0,10\}'
