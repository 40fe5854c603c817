#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for G in ${GROUPS_LIST:-4}; do
  echo "== MMK_WN_GROUPS=$G"
  MMK_WN_GROUPS=$G MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 timeout 300 python bench.py --workload ${WORKLOAD:-wavenet_cfg4} --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps|us_per_ar_step" | sed -e 's/"config".*"ar_steps/"ar_steps/' | cut -c1-900
done
