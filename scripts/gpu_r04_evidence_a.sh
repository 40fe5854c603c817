#!/bin/bash
# round 4 evidence, part A: the whole GPU suite, the default bench line (cfg 4, 32 clips per GPU), the lines for 64 / 128 / 256 clips per GPU,
# the other workloads' lines
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 900 python bench.py > gpurun_out/r04/bench_wavenet_cfg4.json 2> gpurun_out/r04/bench_wavenet_cfg4.err; echo "default bench exit $?"; cut -c1-300 gpurun_out/r04/bench_wavenet_cfg4.json
for c in 64 128 256; do
  timeout 900 python bench.py --clips $c --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04/bench_wavenet_cfg4_clips$c.json 2> gpurun_out/r04/bench_wavenet_cfg4_clips$c.err; echo "clips $c exit $?"
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"us_per_step_in_kernel": [0-9.]*' gpurun_out/r04/bench_wavenet_cfg4_clips$c.json | tr '\n' ' '; echo
done
for WL in wavenet_cfg2 srnn_cfg3 s2s_cfg5 mulaw stft istft gla; do
  timeout 600 python bench.py --workload $WL > gpurun_out/r04/bench_$WL.json 2> gpurun_out/r04/bench_$WL.err; echo "== $WL exit $?"; cut -c1-260 gpurun_out/r04/bench_$WL.json
done
echo "== own launcher on a one-GPU box (must fail loudly)"; python bench.py --gpus 2 --steps 1 --warmup 0 --seconds 0.05 --no-cpu-baseline; echo "exit $?"
echo "== torchrun 1 rank"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 1 --warmup 0 --seconds 0.1 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-300
