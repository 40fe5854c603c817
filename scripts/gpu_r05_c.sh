#!/bin/bash
# round 5: resident SampleRNN kernel - spare CUs 0 (top tier 2 row tiles) against 8 (4 row tiles); parity subset, bench lines, stamps
mkdir -p gpurun_out/r05c
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_networks.py tests/test_gpu_baseline_configs.py -m gpu -q --timeout=300 -x -k "resident or cfg3 or cfg1" > gpurun_out/r05c/pytest_resident.log 2>&1
echo "pytest resident exit: $?" | tee -a gpurun_out/r05c/pytest_resident.log
tail -5 gpurun_out/r05c/pytest_resident.log
for sp in 0 8 0 8; do
timeout 300 python bench.py --tuning MMK_SRNN_SPARE_CUS=$sp --workload srnn_cfg3 --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r05c/bench_srnn_spare$sp.json 2> gpurun_out/r05c/bench_srnn.err; echo "spare $sp: bench exit $?"
grep -o '"value".\{0,30\}\|"us_per_ar_step".\{0,10\}\|"us_per_step".\{0,10\}' gpurun_out/r05c/bench_srnn_spare$sp.json | tr '\n' ' '; echo
done
MMK_DIAG_LIB=1 MMK_SRNN_STAMPS=1 timeout 300 python scripts/srnn_stamps.py 2>&1 | grep "resident kernel" | tee gpurun_out/r05c/srnn_stamps.log
