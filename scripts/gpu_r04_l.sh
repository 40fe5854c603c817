#!/bin/bash
# round 4: free run - every stage at the pace of its own work (diagnostic build, wrong results): the stages' service times, and what
# the two roles take alone (dbg 4 free run, + 8 no bias products, + 16 idle chain waves, + 32 no hidden-sum fetch, + 1 no conditioning row load,
# + 64 no per-phase stamps: the loops as the product build runs them)
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
for cl in 32 64; do
for dbg in 68 76 84 92 100; do
echo "== free run, dbg $dbg, clips $cl"
MMK_WN_SPIPE_DBG=$dbg MMK_WN_STAMP_STAGE=5 MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 timeout 300 python bench.py --clips $cl --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "FREE RUN|rror" | cut -c1-420 | tail -1
done
done > gpurun_out/r04/spipe_freerun3.log 2>&1
cat gpurun_out/r04/spipe_freerun3.log | cut -c1-420
