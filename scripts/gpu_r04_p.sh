#!/bin/bash
# round 4: the stage pipeline on narrower heads / two conditioning inputs, then its other parity tests
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_networks.py -q -x -k "narrower_heads" 2>&1 | tail -25
timeout 1500 python -m pytest tests/test_gpu_networks.py tests/test_gpu_baseline_configs.py -q -x -k "stage_pipeline or cfg4" 2>&1 | tail -5
