#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${WORKLOAD:-wavenet_cfg4}
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${WL}_$C      # a fresh directory per pass: the summary globs whatever lies in it
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${WL}_$C -- python3 $R/scripts/pmc_target.py > $R/gpurun_out/pmc_${WL}_$C.log 2>&1
  echo "pmc $C exit: $?"; tail -2 $R/gpurun_out/pmc_${WL}_$C.log
done
cd $R
python scripts/pmc_summary.py gpurun_out/pmc_${WL}_FETCH_SIZE gpurun_out/pmc_${WL}_fetch_summary.csv | head -12
python scripts/pmc_summary.py gpurun_out/pmc_${WL}_WRITE_SIZE gpurun_out/pmc_${WL}_write_summary.csv | head -12
find gpurun_out -name "*counter_collection.csv" -size +8M -delete
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
