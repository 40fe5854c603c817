#!/bin/bash
# round 4, first pass: cfg-4 parity at 32 / 64 / 128 / 136 clips, bench lines per clip count, A/B of the library variants, stage stamps
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_baseline_configs.py -k "cfg4" -x -q --durations=8 2>&1 | tail -25
for c in 32 64 128 256; do
  timeout 600 python bench.py --clips $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04/bench_cfg4_c$c.json 2> gpurun_out/r04/bench_cfg4_c$c.err; echo "clips $c exit $?"
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"us_per_step_in_kernel": [0-9.]*' gpurun_out/r04/bench_cfg4_c$c.json | tr '\n' ' '; echo
done
bash scripts/gpu_ab.sh 2>&1 | tail -20
STAGES="1 5 20" bash scripts/gpu_spipe_stages.sh > gpurun_out/r04/spipe_stamps.log 2>&1; cut -c1-600 gpurun_out/r04/spipe_stamps.log | tail -20
