#!/bin/bash
# round 5: soak of the batched stage pipeline - 20 passes each at 256 and 128 clips per GPU (16 000 steps a pass); a hand-off that timed out shows as a warning
# ("regenerating this batch") on stderr and in the pass time
mkdir -p gpurun_out/r05k
export TMPDIR=/tmp
for n in 256 128; do
  timeout 1500 python bench.py --workload wavenet_cfg4 --clips $n --steps 20 --warmup 1 --no-cpu-baseline --no-strong-leg > gpurun_out/r05k/soak_$n.json 2> gpurun_out/r05k/soak_$n.err
  echo "clips $n exit $?: $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r05k/soak_$n.json | head -2 | tr '\n' ' ') warnings: $(grep -c "regenerating\|timed out" gpurun_out/r05k/soak_$n.err)"
done
