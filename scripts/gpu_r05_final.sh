#!/bin/bash
# round 5, last check of the tree as the driver will run it: build(), smoke(), the GPU suite, the default bench line
export TMPDIR=/tmp
mkdir -p gpurun_out/r05f
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('build + smoke ok')" 2>&1 | tail -3
timeout 3000 python -m pytest tests -m gpu -q --timeout=900 2>&1 | tail -3
python bench.py 2>/dev/null | tee gpurun_out/r05f/bench_final.json | cut -c1-330
