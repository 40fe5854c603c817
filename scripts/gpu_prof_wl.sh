#!/bin/bash
# rocprofv3 kernel stats of one bench workload: WORKLOAD=<name> [PROF_SECONDS=0.1] [TAG=r04_v1] [EXTRA="--clips 64"]
# Every run writes into a FRESH directory and exactly that run's summary is copied to gpurun_out/${TAG}_${WL}_kernel_stats.csv
# (round 3 collected every CSV that had ever landed in one directory and tracked the wrong one).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
WL=${WORKLOAD:-s2s_cfg5}
TAG=${TAG:-run}
OUT=$R/gpurun_out/prof_${WL}_$(date +%s)_$$
rm -rf $OUT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --workload $WL --steps 1 --warmup 1 --seconds ${PROF_SECONDS:-0.1} --no-cpu-baseline $EXTRA > $OUT.log 2>&1
echo "rocprof exit: $?"; tail -1 $OUT.log | cut -c1-600
cd $R
n=$(find $OUT -name "*kernel_stats.csv" | wc -l)
if [ "$n" != "1" ]; then echo "expected ONE kernel_stats.csv in $OUT, found $n"; fi
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${TAG}_${WL}_kernel_stats.csv
grep '^{' $OUT.log | tail -1 > gpurun_out/${TAG}_${WL}_profiled_bench_line.json
head -8 gpurun_out/${TAG}_${WL}_kernel_stats.csv | cut -c1-200
rm -rf $OUT
