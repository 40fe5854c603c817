#!/bin/bash
# Per-phase wall-clock stamps of the pipelined WaveNet kernel (diagnostic build), one line per stage.
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/pipe_stamps.log
for ST in "0 0" "3 0" "6 0" "7 0" "7 5"; do
  set -- $ST
  echo "== stage $1 owner $2" >> gpurun_out/pipe_stamps.log
  MMK_WN_STAMPS=1 MMK_WN_STAMP_STAGE=$1 MMK_WN_STAMP_OWNER=$2 timeout 300 python bench.py --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps" | tail -1 >> gpurun_out/pipe_stamps.log
done
cat gpurun_out/pipe_stamps.log | cut -c1-1500
