#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_callers.py -m gpu -q --timeout=300 > gpurun_out/pytest_callers.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_callers.log
tail -60 gpurun_out/pytest_callers.log
