#!/bin/bash
# full GPU suite, then bench + rocprof stats for the secondary workloads
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
for WL in s2s_cfg5 srnn_cfg3 wavenet_cfg2; do
  timeout 600 python bench.py --workload $WL --steps 2 --warmup 1 > gpurun_out/bench_$WL.json 2> gpurun_out/bench_$WL.err
  echo "bench $WL exit $?"; cut -c1-200 gpurun_out/bench_$WL.json; grep -o '"roofline".*' gpurun_out/bench_$WL.json | cut -c1-700
done
cd /tmp
for WL in s2s_cfg5 srnn_cfg3; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$WL -- python3 $R/bench.py --workload $WL --steps 1 --warmup 1 --seconds 0.25 --no-cpu-baseline > $R/gpurun_out/prof_$WL.log 2>&1
  echo "rocprof $WL exit: $?"
  for f in $(find $R/gpurun_out/prof_$WL -name "*kernel_stats.csv"); do head -14 $f | cut -c1-160; done
  find $R/gpurun_out/prof_$WL -name "*kernel_trace.csv" -size +20M -delete
done
