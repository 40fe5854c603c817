#!/bin/bash
# round 4, final tree: the path sweep again (the stage pipeline's beat changed), and the PMC passes of the cfg-4 kernel
mkdir -p gpurun_out/r04d
export TMPDIR=/tmp
timeout 1500 python scripts/wn_path_sweep.py --steps 192 --out gpurun_out/r04d/wn_path_sweep.json 2> gpurun_out/r04d/wn_path_sweep.err | tee gpurun_out/r04d/wn_path_sweep.log | grep "^|" | tail -40
WORKLOAD=wavenet_cfg4 bash scripts/gpu_pmc.sh 2>&1 | tail -12 | cut -c1-200
cp gpurun_out/pmc_wavenet_cfg4_fetch_summary.csv gpurun_out/pmc_wavenet_cfg4_write_summary.csv gpurun_out/r04d/ 2>/dev/null
