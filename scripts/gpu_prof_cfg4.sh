#!/bin/bash
# kernel trace + stats of the default bench (wavenet_cfg4): gpurun_out/prof_cfg4/
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_cfg4
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_cfg4 -o cfg4 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_cfg4.log 2>&1
echo "exit $?"
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_cfg4 -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -25 {} | cut -c1-220'
grep -o '"value".\{0,30\}\|"us_per_ar_step".\{0,10\}' gpurun_out/prof_cfg4.log
find gpurun_out/prof_cfg4 -name "*kernel_trace.csv" -size +20M -delete
