#!/bin/bash
# round 5: the all-resident SampleRNN kernel - parity of the SampleRNN tests, then the cfg-3 bench line
mkdir -p gpurun_out/r05a
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_networks.py -m gpu -q --timeout=300 -x -k "resident" > gpurun_out/r05a/pytest_resident.log 2>&1
echo "pytest resident exit: $?" | tee -a gpurun_out/r05a/pytest_resident.log
tail -30 gpurun_out/r05a/pytest_resident.log
timeout 300 python bench.py --workload srnn_cfg3 --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r05a/bench_srnn.json 2> gpurun_out/r05a/bench_srnn.err; echo "bench exit $?"
tail -3 gpurun_out/r05a/bench_srnn.err
grep -o '"value".\{0,30\}\|"us_per_ar_step".\{0,10\}\|"us_per_step".\{0,10\}' gpurun_out/r05a/bench_srnn.json
timeout 1500 python -m pytest tests/test_gpu_networks.py tests/test_gpu_baseline_configs.py tests/test_gpu_callers.py -m gpu -q --timeout=300 -k "sample_rnn or srnn or cfg1 or cfg3 or chunks or callback or from_config or ensemble" > gpurun_out/r05a/pytest_srnn.log 2>&1
echo "pytest srnn exit: $?" | tee -a gpurun_out/r05a/pytest_srnn.log
tail -30 gpurun_out/r05a/pytest_srnn.log
