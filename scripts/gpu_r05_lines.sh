#!/bin/bash
# round 5: bench lines of the other workloads on the round's final tree (cfg 2, SampleRNN cfg 3, Seq2Seq cfg 5, feature kernels) + the 256-clip cfg-4 line
mkdir -p gpurun_out/r05c
export TMPDIR=/tmp
for WL in wavenet_cfg2 srnn_cfg3 s2s_cfg5 mulaw stft istft gla; do
  timeout 900 python bench.py --workload $WL --steps 2 --warmup 1 > gpurun_out/r05c/bench_$WL.json 2> gpurun_out/r05c/bench_$WL.err
  echo "== $WL exit $?: $(grep -o '"value": [0-9.]*\|"frac": [0-9.]*\|"us_per_ar_step": [0-9.]*' gpurun_out/r05c/bench_$WL.json | head -4 | tr '\n' ' ')"
done
timeout 900 python bench.py --workload wavenet_cfg4 --clips 256 --no-cpu-baseline --no-strong-leg > gpurun_out/r05c/bench_wavenet_cfg4_clips256.json 2> gpurun_out/r05c/bench256.err
echo "== clips 256 exit $?: $(grep -o '"value": [0-9.]*\|"traffic": [0-9]*' gpurun_out/r05c/bench_wavenet_cfg4_clips256.json | tr '\n' ' ')"
