#!/bin/bash
# stage pipeline, diagnostic build: the chain's time line with / without the helpers' slow loads
export TMPDIR=/tmp
for dbg in 0 1 2 3; do
echo "== MMK_WN_SPIPE_DBG=$dbg"
MMK_WN_SPIPE_DBG=$dbg MMK_WN_STAMP_STAGE=5 MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 timeout 300 python bench.py --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps|us_per_step_in_kernel" | cut -c1-1500 | grep -v metric | grep "publish time" | cut -c60-400
done
