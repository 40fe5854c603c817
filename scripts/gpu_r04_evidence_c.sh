#!/bin/bash
# round 4 evidence, part C: the ablation of the helper-loop changes (library variants, interleaved, two rounds, 32 and 64 clips) and the
# clips curve of the final build
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
(
EXTRA="--steps 1 --warmup 1 --seconds 0.25" bash scripts/gpu_ab.sh
EXTRA="--clips 64 --steps 1 --warmup 1 --seconds 0.25" bash scripts/gpu_ab.sh | sed 's/^/clips64 /'
) 2>&1 | tee gpurun_out/r04/ablation.log
for c in 8 16 24 28 32 36 40 48 64 96 128; do
  r=$(timeout 300 python bench.py --clips $c --no-cpu-baseline --steps 1 --warmup 1 --seconds 0.128 2>/dev/null | grep -o "\"us_per_step_in_kernel\": [0-9.]*" | grep -o "[0-9.]*$")
  echo "clips $c us_per_step $r"
done 2>&1 | tee gpurun_out/r04/clips_curve.log
