#!/bin/bash
# stage pipeline, diagnostic build: average wait / compute of chain wave 0 at several stages
export TMPDIR=/tmp
for st in ${STAGES:-0 4 10 19 20 21 29}; do
echo "== stage $st"
MMK_WN_STAMP_STAGE=$st MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 timeout 300 python bench.py --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps" | cut -c1-1400 | grep -v "per stage" | grep -v "publish time" | sed "s/.*shader cycles per visit: //; s/.*cycles per iteration: /helper: /"
done
