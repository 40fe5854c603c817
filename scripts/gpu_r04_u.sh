#!/bin/bash
# round 4: SampleRNN / layer-pipeline changes: parity, then the two bench lines
export TMPDIR=/tmp
mkdir -p gpurun_out/r04d
timeout 1800 python -m pytest tests/test_gpu_networks.py tests/test_gpu_baseline_configs.py tests/test_gpu_callers.py -q -x -k "sample_rnn or srnn or cfg1 or cfg3 or cfg2 or layer_pipeline or lpipe or chunks or callback or ensemble or multi_input" 2>&1 | tail -5
for WL in srnn_cfg3 wavenet_cfg2; do
  timeout 600 python bench.py --workload $WL --steps 4 --warmup 1 > gpurun_out/r04d/bench2_$WL.json 2>/dev/null; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"us_per_step": [0-9.]*\|"matches_gpu_output": [a-z]*' gpurun_out/r04d/bench2_$WL.json | tr '\n' ' ' | sed "s/^/$WL: /"; echo
done
