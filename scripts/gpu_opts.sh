#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=300 -k "options or variants or unsupported or stacks" > gpurun_out/pytest_opts.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_opts.log
tail -40 gpurun_out/pytest_opts.log
