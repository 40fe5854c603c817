#!/bin/bash
# round 4 evidence, part B: rocprofv3 kernel stats (each into a fresh directory), PMC traffic passes, the stage pipeline's stamps
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
# cfg 4: the default bench command itself under the profiler
OUT=$R/gpurun_out/prof_cfg4_$$; rm -rf $OUT
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --no-cpu-baseline > $OUT.log 2>&1 ); echo "rocprof cfg4 exit $?"
f=$(find $OUT -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04/wavenet_cfg4_kernel_stats.csv; grep '^{' $OUT.log | tail -1 > gpurun_out/r04/wavenet_cfg4_profiled_bench_line.json; head -6 gpurun_out/r04/wavenet_cfg4_kernel_stats.csv | cut -c1-200; rm -rf $OUT
for WL in wavenet_cfg2 srnn_cfg3 s2s_cfg5; do
  TAG=r04 WORKLOAD=$WL PROF_SECONDS=1 bash scripts/gpu_prof_wl.sh 2>&1 | tail -6 | cut -c1-200
  mv gpurun_out/r04_${WL}_kernel_stats.csv gpurun_out/r04/${WL}_kernel_stats.csv; mv gpurun_out/r04_${WL}_profiled_bench_line.json gpurun_out/r04/${WL}_profiled_bench_line.json
done
WORKLOAD=wavenet_cfg4 bash scripts/gpu_pmc.sh 2>&1 | tail -16 | cut -c1-200
WORKLOAD=wavenet_cfg2 bash scripts/gpu_pmc.sh 2>&1 | tail -16 | cut -c1-200
WORKLOADS="stft istft gla" bash scripts/gpu_pmc_feat.sh 2>&1 | tail -20 | cut -c1-200
cp gpurun_out/pmc_*summary.csv gpurun_out/pmcf_*summary.csv gpurun_out/r04/ 2>/dev/null
for st in 5 20; do
echo "== stage $st clips 32"
MMK_WN_STAMP_STAGE=$st MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 timeout 300 python bench.py --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps" | cut -c1-2400
done > gpurun_out/r04/spipe_stage_latency.log 2>&1
grep -E "==|cycles per|sum" gpurun_out/r04/spipe_stage_latency.log | cut -c1-1200 | head -12
