#!/bin/bash
# round 4: networks of several inputs / targets on the device, then the whole GPU suite
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_networks.py -q -x -k "multi_input" 2>&1 | tail -30
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8
