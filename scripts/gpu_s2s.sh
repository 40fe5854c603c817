#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=300 -k "seq2seq" > gpurun_out/pytest_s2s.log 2>&1
echo "pytest exit: $?"; grep -v "^$" gpurun_out/pytest_s2s.log | tail -15 | cut -c1-200
for F in 0 1; do
  echo "== MMK_S2S_FUSED=$F"
  timeout 600 python bench.py --tuning MMK_S2S_FUSED=$F --workload s2s_cfg5 --steps 2 --warmup 1 ${BENCH_ARGS} 2>&1 | tail -1 | cut -c1-300
done
