#!/bin/bash
# HBM traffic counters (separate --pmc passes, kernel-trace only) of the spectral feature kernels: WORKLOADS="stft istft gla"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for WL in ${WORKLOADS:-stft istft gla}; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmcf_${WL}_$C      # a fresh directory per pass: the summary globs whatever lies in it
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcf_${WL}_$C -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcf_${WL}_$C.log 2>&1
    echo "pmc $WL $C exit: $?"
  done
done
cd $R
for WL in ${WORKLOADS:-stft istft gla}; do
  python scripts/pmc_summary.py gpurun_out/pmcf_${WL}_FETCH_SIZE gpurun_out/pmcf_${WL}_fetch_summary.csv | grep -E "mmk|kernel," | head -4
  python scripts/pmc_summary.py gpurun_out/pmcf_${WL}_WRITE_SIZE gpurun_out/pmcf_${WL}_write_summary.csv | grep -E "mmk|kernel," | head -4
done
find gpurun_out -name "*counter_collection.csv" -size +8M -delete
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
