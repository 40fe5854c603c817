#!/bin/bash
# HBM traffic counters of the Seq2Seq cfg-5 kernels (separate passes per counter, as the guide prescribes)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmcs_s2s_$C      # a fresh directory per pass: the summary globs whatever lies in it
  timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_s2s_$C -- python3 $R/bench.py --workload s2s_cfg5 --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcs_s2s_$C.log 2>&1 < /dev/null
  echo "pmc s2s $C exit: $?"
done
cd $R
python scripts/pmc_summary.py gpurun_out/pmcs_s2s_FETCH_SIZE gpurun_out/pmcs_s2s_fetch_summary.csv | grep -E "mmk|kernel,|anonymous" | head -14
python scripts/pmc_summary.py gpurun_out/pmcs_s2s_WRITE_SIZE gpurun_out/pmcs_s2s_write_summary.csv | grep -E "mmk|kernel,|anonymous" | head -14
find gpurun_out -name "*counter_collection.csv" -size +8M -delete
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
exit 0
