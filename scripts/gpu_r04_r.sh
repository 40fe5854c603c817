#!/bin/bash
# round 4: what the chain waves' hidden-unit hand-over costs a step (diagnostic build, timing only: wrong results)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
for cl in 32 64; do
for dbg in 64 192 448; do
r=$(MMK_WN_SPIPE_DBG=$dbg MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 timeout 300 python bench.py --clips $cl --steps 1 --warmup 1 --seconds 0.128 --no-cpu-baseline 2>/dev/null | grep -o '"us_per_step_in_kernel": [0-9.]*')
echo "clips $cl dbg $dbg $r"
done
done 2>&1 | tee gpurun_out/r04/hid_cost.log
