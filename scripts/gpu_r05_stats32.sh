#!/bin/bash
# round 5: kernel stats and HBM counters of the driver's default command (cfg 4, 32 clips: wavenet_spipe_kernel) on the round's final tree
mkdir -p gpurun_out/r05s
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
rm -rf $R/gpurun_out/r05s/stats
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05s/stats -- python3 $R/bench.py --no-cpu-baseline --no-strong-leg > $R/gpurun_out/r05s/profiled_bench_line.json 2> $R/gpurun_out/r05s/stats.err
echo "stats exit $?"
f=$(find $R/gpurun_out/r05s/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r05s/wavenet_cfg4_kernel_stats.csv && head -4 $f | cut -c1-220
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/r05s/pmc_$C
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/r05s/pmc_$C -- python3 $R/scripts/pmc_target.py > $R/gpurun_out/r05s/pmc_$C.log 2>&1
  echo "pmc $C exit: $?"
done
cd $R
python scripts/pmc_summary.py gpurun_out/r05s/pmc_FETCH_SIZE gpurun_out/r05s/pmc_fetch_size_summary.csv | grep -E "spipe|kernel," | head -3
python scripts/pmc_summary.py gpurun_out/r05s/pmc_WRITE_SIZE gpurun_out/r05s/pmc_write_size_summary.csv | grep -E "spipe|kernel," | head -3
find gpurun_out/r05s -name "*counter_collection.csv" -size +4M -delete
find gpurun_out/r05s -name "*kernel_trace.csv" -size +4M -delete
rm -rf gpurun_out/r05s/stats
