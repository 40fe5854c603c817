#!/bin/bash
# round 2, first call: the whole GPU suite + smoke + the default bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" >> gpurun_out/smoke.log
tail -3 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench_wavenet_cfg4.json 2> gpurun_out/bench_wavenet_cfg4.err; echo "bench exit $?"
cat gpurun_out/bench_wavenet_cfg4.json | cut -c1-1500
