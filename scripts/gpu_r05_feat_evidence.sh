#!/bin/bash
# round 5: feature-kernel evidence - parity, bench lines (with the CPU baseline), SQ counters (LDS bank conflicts) and HBM traffic counters
V=${V:-r05_v1}
mkdir -p gpurun_out/$V
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_gpu_features.py -m gpu -q --timeout=300 > gpurun_out/$V/pytest_feat.log 2>&1
echo "pytest exit: $?"; tail -2 gpurun_out/$V/pytest_feat.log
for WL in mulaw stft istft gla; do
  timeout 600 python bench.py --workload $WL --steps 5 --warmup 2 > gpurun_out/$V/${V}_bench_$WL.json 2>/dev/null
  python -c "import sys,json; d=json.loads(open('gpurun_out/$V/${V}_bench_$WL.json').read()); print(d['config']['workload'][:34], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done
cd /tmp
for WL in stft istft gla; do
  rm -rf $R/gpurun_out/$V/pmcq_$WL $R/gpurun_out/$V/pmcq2_$WL
  timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/$V/pmcq_$WL -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/$V/pmcq_$WL.log 2>&1
  echo "pmc $WL exit: $?"
  timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/$V/pmcq2_$WL -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/$V/pmcq2_$WL.log 2>&1
  echo "pmc2 $WL exit: $?"
done
for WL in mulaw stft istft gla; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/$V/pmcf_${WL}_$C
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/$V/pmcf_${WL}_$C -- python3 $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/$V/pmcf_${WL}_$C.log 2>&1
    echo "pmc $WL $C exit: $?"
  done
done
cd $R
for WL in stft istft gla; do
  python scripts/pmc_summary.py gpurun_out/$V/pmcq_$WL gpurun_out/$V/${V}_pmc_sq_${WL}_a.csv | grep -E "stft|gla|kernel," | head -9
  python scripts/pmc_summary.py gpurun_out/$V/pmcq2_$WL gpurun_out/$V/${V}_pmc_sq_${WL}_b.csv | grep -E "stft|gla|kernel," | head -9
  cat gpurun_out/$V/${V}_pmc_sq_${WL}_a.csv > gpurun_out/$V/${V}_pmc_sq_${WL}_summary.csv; tail -n +2 gpurun_out/$V/${V}_pmc_sq_${WL}_b.csv >> gpurun_out/$V/${V}_pmc_sq_${WL}_summary.csv
done
for WL in mulaw stft istft gla; do
  python scripts/pmc_summary.py gpurun_out/$V/pmcf_${WL}_FETCH_SIZE gpurun_out/$V/${V}_pmc_${WL}_fetch_size_summary.csv | grep -E "mmk|kernel," | head -3
  python scripts/pmc_summary.py gpurun_out/$V/pmcf_${WL}_WRITE_SIZE gpurun_out/$V/${V}_pmc_${WL}_write_size_summary.csv | grep -E "mmk|kernel," | head -3
done
find gpurun_out/$V -name "*counter_collection.csv" -size +2M -delete
find gpurun_out/$V -name "*kernel_trace.csv" -size +2M -delete
