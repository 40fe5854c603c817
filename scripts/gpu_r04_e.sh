#!/bin/bash
# round 4: us per AR step of cfg 4 against the number of clips in the ring (latency-bound below the knee, beat-bound above), and stage stamps
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
for c in ${CLIPS:-4 8 16 24 32 40 48 64}; do
  r=$(timeout 300 python bench.py --clips $c --steps 1 --warmup 1 --seconds 0.25 --no-cpu-baseline 2>/dev/null | grep -o '"us_per_step_in_kernel": [0-9.]*')
  echo "clips $c $r"
done | tee gpurun_out/r04/clips_curve.log
for st in ${STAGES:-3 4}; do
for c in ${STAMP_CLIPS:-32 64}; do
echo "== stage $st clips $c"
MMK_WN_STAMP_STAGE=$st MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 timeout 300 python bench.py --clips $c --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps" | cut -c1-2400
done
done > gpurun_out/r04/spipe_stamps_full2.log 2>&1
grep -E "==|cycles per|sum" gpurun_out/r04/spipe_stamps_full2.log | cut -c1-1300
