#!/bin/bash
# rocprofv3 kernel stats of one bench workload: WORKLOAD=<name> [PROF_SECONDS=0.1]
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${WORKLOAD:-s2s_cfg5}
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$WL -- python3 $R/bench.py --workload $WL --steps 1 --warmup 1 --seconds ${PROF_SECONDS:-0.1} --no-cpu-baseline > $R/gpurun_out/prof_$WL.log 2>&1
echo "rocprof exit: $?"; tail -1 $R/gpurun_out/prof_$WL.log | cut -c1-600
cd $R
for f in $(find gpurun_out/prof_$WL -name "*kernel_stats.csv"); do head -12 $f | cut -c1-200; done
find gpurun_out/prof_$WL -name "*kernel_trace.csv" -size +20M -delete
