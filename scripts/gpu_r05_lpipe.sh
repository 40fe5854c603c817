#!/bin/bash
# round 5: conditioned networks through the layer pipeline - the path sweep rows for 64 channels (1024 steps per call), BASELINE config 2's bench line
mkdir -p gpurun_out/r05l
export TMPDIR=/tmp
timeout 600 python scripts/wn_path_sweep.py --steps 1024 --channels 64 --blocks "10" --clips 8,32 --out gpurun_out/r05l/wn_path_sweep_64_1024.json > gpurun_out/r05l/wn_path_sweep_64_1024.md 2> gpurun_out/r05l/sweep.err
echo "sweep exit $?"; grep "^|" gpurun_out/r05l/wn_path_sweep_64_1024.md
for i in 1 2; do
timeout 300 python bench.py --workload wavenet_cfg2 --no-cpu-baseline > gpurun_out/r05l/bench_cfg2_$i.json 2> gpurun_out/r05l/bench.err; echo "bench exit $?"
grep -o '"us_per_step_in_kernel": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/r05l/bench_cfg2_$i.json | tr '\n' ' '; echo
done
