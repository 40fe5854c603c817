#!/bin/bash
# round 4, last check of the tree as the driver will run it: build(), smoke(), the GPU suite, the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out/r04d
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build + smoke ok')" 2>&1 | tail -3
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3
python bench.py 2>/dev/null | tee gpurun_out/r04d/bench_final.json | cut -c1-330
