#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_srnn -- python3 $R/bench.py --workload srnn_cfg3 --steps 1 --warmup 1 --seconds 0.1 --no-cpu-baseline > $R/gpurun_out/prof_srnn.log 2>&1
echo "rocprof exit: $?"
cd $R
for f in $(find gpurun_out/prof_srnn -name "*kernel_stats.csv"); do head -14 $f | cut -c1-200; done
find gpurun_out/prof_srnn -name "*kernel_trace.csv" -size +20M -delete
