#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=300 -k "sample_rnn" > gpurun_out/pytest_srnn.log 2>&1
echo "pytest exit: $?"; grep -v "^$" gpurun_out/pytest_srnn.log | tail -25 | cut -c1-200
for F in 0 1; do
  echo "== MMK_SRNN_FUSED=$F"
  MMK_SRNN_FUSED=$F timeout 600 python bench.py --workload srnn_cfg3 --steps 2 --warmup 1 ${BENCH_ARGS} 2>&1 | tail -1 | cut -c1-400
done
