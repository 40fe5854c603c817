#!/bin/bash
# Where do the spectral kernels' wave-cycles go?  One rocprofv3 --pmc pass of SQ counters per workload (kernel-trace only).
#   WORKLOADS="stft istft gla" bash scripts/gpu_pmc_sq.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for WL in ${WORKLOADS:-stft istft}; do
  rm -rf $R/gpurun_out/pmcq_$WL      # a fresh directory per pass: the summary globs whatever lies in it
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmcq_$WL -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcq_$WL.log 2>&1
  echo "pmc $WL exit: $?"
  rm -rf $R/gpurun_out/pmcq2_$WL      # a fresh directory per pass: the summary globs whatever lies in it
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmcq2_$WL -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmcq2_$WL.log 2>&1
  echo "pmc2 $WL exit: $?"
done
cd $R
for WL in ${WORKLOADS:-stft istft}; do
  python scripts/pmc_summary.py gpurun_out/pmcq_$WL gpurun_out/pmcq_${WL}_summary.csv | grep -E "stft|gla|kernel," | head -12
  python scripts/pmc_summary.py gpurun_out/pmcq2_$WL gpurun_out/pmcq2_${WL}_summary.csv | grep -E "stft|gla|kernel," | head -12
done
find gpurun_out -name "*counter_collection.csv" -size +8M -delete
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete
