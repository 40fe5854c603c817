#!/bin/bash
# cfg4 headline: bench line, rocprofv3 kernel stats of the same command, PMC HBM traffic (separate passes) and MFMA-busy
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python bench.py > gpurun_out/bench_wavenet_cfg4.json 2> gpurun_out/bench_wavenet_cfg4.err; echo "bench exit $?"
cut -c1-300 gpurun_out/bench_wavenet_cfg4.json; grep -o '"roofline".*' gpurun_out/bench_wavenet_cfg4.json | cut -c1-900
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_wavenet_cfg4 -- python3 $R/bench.py --steps 1 --warmup 1 --seconds 0.25 --no-cpu-baseline > $R/gpurun_out/prof_wavenet_cfg4.log 2>&1
echo "rocprof exit: $?"
for f in $(find $R/gpurun_out/prof_wavenet_cfg4 -name "*kernel_stats.csv"); do head -8 $f | cut -c1-200; done
find $R/gpurun_out/prof_wavenet_cfg4 -name "*kernel_trace.csv" -size +20M -delete
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wavenet_cfg4_$C -- python3 $R/scripts/pmc_target.py > $R/gpurun_out/pmc_wavenet_cfg4_$C.log 2>&1
  echo "pmc $C exit: $?"; tail -1 $R/gpurun_out/pmc_wavenet_cfg4_$C.log
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma_wavenet -- python3 $R/scripts/pmc_target.py > $R/gpurun_out/pmc_mfma_wavenet.log 2>&1
echo "mfma exit: $?"
cd $R
python scripts/pmc_summary.py gpurun_out/pmc_wavenet_cfg4_FETCH_SIZE gpurun_out/pmc_wavenet_cfg4_fetch_summary.csv | head -6
python scripts/pmc_summary.py gpurun_out/pmc_wavenet_cfg4_WRITE_SIZE gpurun_out/pmc_wavenet_cfg4_write_summary.csv | head -6
python scripts/pmc_summary.py gpurun_out/pmc_mfma_wavenet gpurun_out/pmc_mfma_wavenet_summary.csv | head -8
find gpurun_out -name "*counter_collection.csv" -size +8M -delete
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
