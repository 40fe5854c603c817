#!/bin/bash
export TMPDIR=/tmp
for ST in "7 0"; do
  set -- $ST
  echo "== stage $1 owner $2"
  MMK_WN_STAMPS=1 MMK_WN_STAMP_STAGE=$1 MMK_WN_STAMP_OWNER=$2 timeout 300 python bench.py --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps" | tail -1 | cut -c150-1300
done
