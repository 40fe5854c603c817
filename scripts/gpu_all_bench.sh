#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for WL in wavenet_cfg2 srnn_cfg3 s2s_cfg5 mulaw stft; do
  timeout 600 python bench.py --workload $WL --steps 2 --warmup 1 > gpurun_out/bench_$WL.json 2> gpurun_out/bench_$WL.err
  echo "== $WL exit $?"; tail -2 gpurun_out/bench_$WL.err | cut -c1-300; cut -c1-1500 gpurun_out/bench_$WL.json
done
echo "== torchrun 1 rank"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 1 --warmup 0 --seconds 0.1 --no-cpu-baseline 2>&1 | tail -3 | cut -c1-600
