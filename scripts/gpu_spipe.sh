#!/bin/bash
# stage-pipeline kernel: parity tests, then the cfg-4 bench line and the stamps of one stage
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_networks.py -x -q -m gpu -k "stage_pipeline" 2>&1 | tail -15
timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 1 --seconds 0.25 > gpurun_out/bench_wn.json 2> gpurun_out/bench_wn.err; echo "bench exit $?"
tail -3 gpurun_out/bench_wn.err
grep -o '"value".\{0,30\}\|"us_per_ar_step".\{0,10\}\|"us_per_step_in_kernel".\{0,10\}\|"mode".\{0,20\}' gpurun_out/bench_wn.json
for dbg in 0; do
MMK_WN_SPIPE_DBG=$dbg MMK_WN_STAMP_STAGE=5 MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 timeout 300 python bench.py --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps" | cut -c1-1500 | tail -3
done
