#!/bin/bash
# bench + rocprofv3 kernel-trace stats of the same command (shorter clip for the trace)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${WORKLOAD:-wavenet_cfg4}
timeout 900 python bench.py --workload $WL --steps ${STEPS:-2} --warmup 1 > gpurun_out/bench_$WL.json 2> gpurun_out/bench_$WL.err
echo "bench exit: $?"; tail -3 gpurun_out/bench_$WL.err; cat gpurun_out/bench_$WL.json
if [ -n "$PROFILE" ]; then
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$WL -- python3 $R/bench.py --workload $WL --steps 1 --warmup 1 --seconds ${PROF_SECONDS:-0.1} --no-cpu-baseline > $R/gpurun_out/prof_$WL.log 2>&1
  echo "rocprof exit: $?"
  cd $R
  find gpurun_out/prof_$WL -name "*stats*" | head
  for f in $(find gpurun_out/prof_$WL -name "*kernel_stats.csv"); do head -20 $f; done
  # keep the merge small: drop the per-dispatch trace, keep the stats
  find gpurun_out/prof_$WL -name "*kernel_trace.csv" -size +20M -delete
fi
