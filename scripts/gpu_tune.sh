#!/bin/bash
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" MMK_WN_STAMPS=1 timeout 300 python bench.py --workload wavenet_cfg4 --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps|us_per_step_in_kernel" | sed -e 's/.*"us_per_step_in_kernel"/us_per_step_in_kernel/' | cut -c1-420 | tail -2; }
run MMK_WN_CPW=2
run MMK_WN_CPW=4
run MMK_WN_CPW=2 MMK_WN_POLL_SLEEP=0
run MMK_WN_CPW=4 MMK_WN_POLL_SLEEP=0
