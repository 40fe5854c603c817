#!/bin/bash
# round 4: which stage of the ring is the slow one?  Beat-bound (64 clips): the chain wave of the slowest stage never waits for a message
mkdir -p gpurun_out/r04
export TMPDIR=/tmp

for st in 0 1 2 3 4 5 8 9 10 11 12 16 19 20 21 24 28 29; do
echo "== stage $st clips 64"
MMK_WN_STAMP_STAGE=$st MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 timeout 300 python bench.py --clips 64 --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps\] (last stage|helper wave|head,)|us_per_ar_step" | cut -c1-700
done > gpurun_out/r04/spipe_stage_sweep64.log 2>&1
grep -E "==|chain wave|head," gpurun_out/r04/spipe_stage_sweep64.log | cut -c1-400
