#!/bin/bash
# round 4: the whole GPU suite after the selection fix (no -x), then the default bench line
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8
python bench.py 2>/dev/null | tee gpurun_out/r04/bench_i.json
