#!/bin/bash
# HBM traffic counters of the SampleRNN cfg-3 generate block: ONE resident launch (srnn_resident_kernel), so counter collection sees a single dispatch;
# 1600 steps.  TUNING (optional) travels through bench.py --tuning (the library reads no environment variable).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${TUNING:+--tuning $TUNING}
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmcs_srnn_$C      # a fresh directory per pass: the summary globs whatever lies in it
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_srnn_$C -- python3 $R/bench.py $T --workload srnn_cfg3 --steps 1 --warmup 0 --seconds 0.1 --no-cpu-baseline > $R/gpurun_out/pmcs_srnn_$C.log 2>&1
  echo "pmc srnn $C exit: $?"
done
cd $R
python scripts/pmc_summary.py gpurun_out/pmcs_srnn_FETCH_SIZE gpurun_out/pmcs_srnn_fetch_summary.csv | grep -E "mmk|kernel," | head -8
python scripts/pmc_summary.py gpurun_out/pmcs_srnn_WRITE_SIZE gpurun_out/pmcs_srnn_write_summary.csv | grep -E "mmk|kernel," | head -8
find gpurun_out -name "*counter_collection.csv" -size +8M -delete
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
