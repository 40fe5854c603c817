#!/bin/bash
# round 4: soak of the stage pipeline's new modes - many passes at several clip counts; a timed-out hand-off shows as a warning
# ("regenerating this batch") and as a collapse of the samples/s
export TMPDIR=/tmp
mkdir -p gpurun_out/r04d
for c in 32 24 40 64 100 128 33; do
  timeout 900 python bench.py --clips $c --steps 12 --warmup 1 --no-cpu-baseline 2> gpurun_out/r04d/soak_$c.err | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ' | sed "s/^/clips $c: /"; echo " warnings: $(grep -c -i 'regenerat\|timed' gpurun_out/r04d/soak_$c.err)"
done 2>&1 | tee gpurun_out/r04d/soak.log
