#!/bin/bash
# kernel stats of the cfg-5 Seq2Seq bench: gpurun_out/prof_s2s/
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_s2s
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o s2s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload s2s_cfg5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1 < /dev/null
echo "exit $?"
grep -o '"value".\{0,30\}\|"us_per_generate_step".\{0,10\}' $OUT/bench.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f" | cut -c1-200
find $OUT -name "*kernel_trace.csv" -size +20M -delete
exit 0
