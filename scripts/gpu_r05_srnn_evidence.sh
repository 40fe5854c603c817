#!/bin/bash
# round 5: SampleRNN evidence - every SampleRNN GPU test, the cfg-3 bench line, the kernel trace and the two PMC passes of the same command
V=${V:-r05_v1}
mkdir -p gpurun_out/$V
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests/test_gpu_networks.py tests/test_gpu_baseline_configs.py tests/test_gpu_callers.py -m gpu -q --timeout=300 -k "sample_rnn or srnn or cfg1 or cfg3 or chunks or callback or from_config or ensemble" > gpurun_out/$V/pytest_srnn.log 2>&1
echo "pytest srnn exit: $?" | tee -a gpurun_out/$V/pytest_srnn.log
tail -4 gpurun_out/$V/pytest_srnn.log
timeout 600 python bench.py --workload srnn_cfg3 --steps 4 --warmup 1 > gpurun_out/$V/${V}_bench_srnn_cfg3.json 2> gpurun_out/$V/bench_srnn.err; echo "bench exit $?"
grep -o '"value".\{0,30\}\|"us_per_ar_step".\{0,10\}\|"us_per_step".\{0,10\}\|"matches_gpu_output".\{0,8\}' gpurun_out/$V/${V}_bench_srnn_cfg3.json | tr '\n' ' '; echo
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$V/prof_srnn -- python3 $R/bench.py --workload srnn_cfg3 --steps 1 --warmup 1 --seconds 0.1 --no-cpu-baseline > $R/gpurun_out/$V/prof_srnn.log 2>&1
echo "rocprof exit: $?"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/$V/pmcs_srnn_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/$V/pmcs_srnn_$C -- python3 $R/bench.py --workload srnn_cfg3 --steps 1 --warmup 0 --seconds 0.1 --no-cpu-baseline > $R/gpurun_out/$V/pmcs_srnn_$C.log 2>&1
  echo "pmc srnn $C exit: $?"
done
cd $R
for f in $(find gpurun_out/$V/prof_srnn -name "*kernel_stats.csv"); do cp $f gpurun_out/$V/${V}_srnn_cfg3_kernel_stats.csv; head -8 $f | cut -c1-220; done
python scripts/pmc_summary.py gpurun_out/$V/pmcs_srnn_FETCH_SIZE gpurun_out/$V/${V}_pmc_srnn_cfg3_fetch_size_summary.csv | grep -E "srnn|kernel," | head -6
python scripts/pmc_summary.py gpurun_out/$V/pmcs_srnn_WRITE_SIZE gpurun_out/$V/${V}_pmc_srnn_cfg3_write_size_summary.csv | grep -E "srnn|kernel," | head -6
find gpurun_out/$V -name "*counter_collection.csv" -size +4M -delete
find gpurun_out/$V -name "*kernel_trace.csv" -size +4M -delete
