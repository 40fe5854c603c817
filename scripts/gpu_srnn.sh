#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_networks.py tests/test_gpu_baseline_configs.py tests/test_gpu_callers.py -m gpu -q --timeout=300 -x -k "sample_rnn or srnn or cfg1 or cfg3 or chunks or callback or from_config or ensemble" > gpurun_out/pytest_srnn.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_srnn.log
tail -25 gpurun_out/pytest_srnn.log
timeout 300 python bench.py --workload srnn_cfg3 --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/bench_srnn.json 2> gpurun_out/bench_srnn.err; echo "bench exit $?"
grep -o '"value".\{0,30\}\|"us_per_ar_step".\{0,10\}\|"us_per_step".\{0,10\}' gpurun_out/bench_srnn.json
timeout 300 python bench.py --tuning MMK_SRNN_RESIDENT=0 --workload srnn_cfg3 --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | grep -o '"value".\{0,30\}\|"us_per_step".\{0,10\}'
