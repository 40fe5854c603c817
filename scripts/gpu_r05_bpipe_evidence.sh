#!/bin/bash
# round 5: cfg 4 beyond the one-clip ring - bench lines at 64 / 128 / 256 / 512 clips per GPU, the driver's default line (32 clips + the strong-scaling leg),
# kernel stats and the HBM counters of the 256-clip launch (wavenet_bpipe_kernel)
mkdir -p gpurun_out/r05b
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for n in 64 128 256 512; do
  timeout 900 python bench.py --workload wavenet_cfg4 --clips $n --no-cpu-baseline --no-strong-leg > gpurun_out/r05b/bench_wavenet_cfg4_clips$n.json 2> gpurun_out/r05b/bench.err
  echo "clips $n exit $?: $(grep -o '"value": [0-9.]*\|"us_per_ar_step": [0-9.]*\|"us_per_step_in_kernel": [0-9.]*' gpurun_out/r05b/bench_wavenet_cfg4_clips$n.json | tr '\n' ' ')"
done
timeout 900 python bench.py > gpurun_out/r05b/bench_wavenet_cfg4.json 2> gpurun_out/r05b/bench_default.err
echo "default exit $?: $(grep -o '"value": [0-9.]*\|"us_per_ar_step": [0-9.]*\|"strong_scaling": {[^}]*}' gpurun_out/r05b/bench_wavenet_cfg4.json | tr '\n' ' ')"
cd /tmp
rm -rf $R/gpurun_out/r05b/stats256
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05b/stats256 -- python3 $R/bench.py --workload wavenet_cfg4 --clips 256 --seconds 0.128 --no-cpu-baseline --no-strong-leg > $R/gpurun_out/r05b/stats256.log 2>&1
echo "stats exit $?"
f=$(find $R/gpurun_out/r05b/stats256 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r05b/wavenet_cfg4_clips256_kernel_stats.csv && head -4 $f | cut -c1-200
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/r05b/pmc_$C
  CLIPS=256 timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/r05b/pmc_$C -- python3 $R/scripts/pmc_target.py > $R/gpurun_out/r05b/pmc_$C.log 2>&1
  echo "pmc $C exit: $?"
done
cd $R
python scripts/pmc_summary.py gpurun_out/r05b/pmc_FETCH_SIZE gpurun_out/r05b/pmc_fetch_size_clips256_summary.csv | grep -E "bpipe|kernel," | head -4
python scripts/pmc_summary.py gpurun_out/r05b/pmc_WRITE_SIZE gpurun_out/r05b/pmc_write_size_clips256_summary.csv | grep -E "bpipe|kernel," | head -4
find gpurun_out/r05b -name "*counter_collection.csv" -size +4M -delete
find gpurun_out/r05b -name "*kernel_trace.csv" -size +4M -delete
rm -rf gpurun_out/r05b/stats256
