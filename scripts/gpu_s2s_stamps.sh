#!/bin/bash
# phase stamps of the resident bi-LSTM kernel (diagnostic build), cfg 5: last launch of the bench's last pass (a decoder layer)
export TMPDIR=/tmp
mkdir -p gpurun_out
MMK_DIAG_LIB=1 MMK_S2S_STAMPS=1 MMK_S2S_STAMP_WG=${1:-0} timeout 300 python bench.py --workload s2s_cfg5 --steps 1 --warmup 0 --no-cpu-baseline 2> gpurun_out/s2s_stamps.err | tail -1 | cut -c1-200
grep "mmk stamps" gpurun_out/s2s_stamps.err | tail -18 > gpurun_out/s2s_stamps.log
cat gpurun_out/s2s_stamps.log | cut -c1-260
