#!/bin/bash
# round 3, cfg 4 (stage pipeline): the bench line, rocprofv3 kernel stats of the same command, PMC traffic passes, diagnostic stamps
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 600 python bench.py > gpurun_out/r03/bench_wavenet_cfg4.json 2> gpurun_out/r03/bench_wavenet_cfg4.err; echo "bench exit $?"
cut -c1-400 gpurun_out/r03/bench_wavenet_cfg4.json
cd /tmp
rm -rf $R/gpurun_out/r03/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r03/prof -o cfg4 --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r03/prof.log 2>&1; echo "rocprof exit $?"
cd $R
head -8 gpurun_out/r03/prof/cfg4_kernel_stats.csv | cut -c1-200
find gpurun_out/r03 -name "*kernel_trace.csv" -delete
WORKLOAD=wavenet_cfg4 bash scripts/gpu_pmc.sh 2>&1 | tail -30
STAGES="1 5 19" bash scripts/gpu_spipe_stages.sh > gpurun_out/r03/spipe_stamps.log 2>&1; tail -12 gpurun_out/r03/spipe_stamps.log | cut -c1-300
MMK_DIAG_LIB=1 MMK_WN_STAMP_STAGE=5 MMK_WN_STAMPS=1 timeout 300 python bench.py --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps" > gpurun_out/r03/spipe_timeline.log
