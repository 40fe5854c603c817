#!/bin/bash
# round 5: cfg 4 bench lines at 64 / 128 / 256 / 512 clips per GPU (default --seconds), the driver's default line (32 clips + strong-scaling leg)
mkdir -p gpurun_out/r05b
export TMPDIR=/tmp
for n in 64 128 256 512; do
  timeout 900 python bench.py --workload wavenet_cfg4 --clips $n --no-cpu-baseline --no-strong-leg > gpurun_out/r05b/bench_wavenet_cfg4_clips$n.json 2> gpurun_out/r05b/bench.err
  echo "clips $n exit $?: $(grep -o '"value": [0-9.]*\|"us_per_ar_step": [0-9.]*\|"us_per_step_in_kernel": [0-9.]*' gpurun_out/r05b/bench_wavenet_cfg4_clips$n.json | tr '\n' ' ')"
done
timeout 900 python bench.py > gpurun_out/r05b/bench_wavenet_cfg4.json 2> gpurun_out/r05b/bench_default.err
echo "default exit $?: $(grep -o '"value": [0-9.]*\|"us_per_ar_step": [0-9.]*\|"strong_scaling": {[^}]*}' gpurun_out/r05b/bench_wavenet_cfg4.json | tr '\n' ' ')"
