#!/bin/bash
# round 4: the padded heads of the layer pipeline / stage pipeline, then the WaveNet parity tests
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_networks.py -q -x -k "narrower_heads" 2>&1 | tail -25
timeout 1800 python -m pytest tests/test_gpu_networks.py tests/test_gpu_baseline_configs.py -q -x -k "layer_pipeline or cfg2 or wavenet" 2>&1 | tail -5
