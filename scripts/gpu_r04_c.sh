#!/bin/bash
# round 4: the whole GPU suite, bench lines per clip count, A/B of the library variants, full stage stamps
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | tail -8
for c in ${CLIPS:-32 64 128}; do
  timeout 600 python bench.py --clips $c --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/r04/bench_cfg4_c$c.json 2> gpurun_out/r04/bench_cfg4_c$c.err; echo "clips $c exit $?"
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"us_per_step_in_kernel": [0-9.]*' gpurun_out/r04/bench_cfg4_c$c.json | tr '\n' ' '; echo
done
bash scripts/gpu_ab.sh 2>&1 | tail -20
EXTRA="--steps 1 --warmup 1 --seconds 0.25 --clips 64" bash scripts/gpu_ab.sh 2>&1 | tail -20
for st in ${STAGES:-5 20}; do
for c in 32 64; do
echo "== stage $st clips $c"
MMK_WN_STAMP_STAGE=$st MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 timeout 300 python bench.py --clips $c --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps" | cut -c1-2400
done
done > gpurun_out/r04/spipe_stamps_full.log 2>&1
grep -E "==|cycles per|sum" gpurun_out/r04/spipe_stamps_full.log | cut -c1-1300
