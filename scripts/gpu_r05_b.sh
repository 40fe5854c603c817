#!/bin/bash
# round 5: resident SampleRNN kernel - quick parity subset, bench line, stamps
mkdir -p gpurun_out/r05b
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_networks.py tests/test_gpu_baseline_configs.py -m gpu -q --timeout=300 -x -k "resident or cfg3 or cfg1" > gpurun_out/r05b/pytest_resident.log 2>&1
echo "pytest resident exit: $?" | tee -a gpurun_out/r05b/pytest_resident.log
tail -5 gpurun_out/r05b/pytest_resident.log
for i in 1 2; do
timeout 300 python bench.py --workload srnn_cfg3 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r05b/bench_srnn.json 2> gpurun_out/r05b/bench_srnn.err; echo "bench exit $?"
grep -o '"value".\{0,30\}\|"us_per_ar_step".\{0,10\}\|"us_per_step".\{0,10\}' gpurun_out/r05b/bench_srnn.json
done
MMK_DIAG_LIB=1 MMK_SRNN_STAMPS=1 timeout 300 python scripts/srnn_stamps.py 2>&1 | grep "resident kernel" | tee gpurun_out/r05b/srnn_stamps.log
