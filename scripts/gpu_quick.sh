#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=300 -k "wavenet" ${PYTEST_ARGS} --durations=5 > gpurun_out/pytest_quick.log 2>&1
echo "pytest exit: $?"; grep -v "^E  \|^    \|^$" gpurun_out/pytest_quick.log | tail -25
