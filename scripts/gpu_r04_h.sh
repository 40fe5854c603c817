#!/bin/bash
# round 4: the whole GPU suite after the selection change, then the sweep's default column again
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
timeout 1500 python scripts/wn_path_sweep.py --steps 192 --out gpurun_out/r04/wn_path_sweep2.json 2> gpurun_out/r04/wn_path_sweep2.err | tee gpurun_out/r04/wn_path_sweep2.log | grep "^|" | tail -40
