#!/bin/bash
# round 3: bench lines of every workload + feature parity tests (gpurun_out/r03/)
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_features.py -x -q -m gpu 2>&1 | tail -3
for wl in ${WORKLOADS:-mulaw stft istft gla wavenet_cfg2 srnn_cfg3 s2s_cfg5}; do
  timeout 600 python bench.py --workload $wl > gpurun_out/r03/bench_$wl.json 2> gpurun_out/r03/bench_$wl.err; echo "$wl exit $?"
  python - <<PY
import json
l=json.load(open("gpurun_out/r03/bench_$wl.json"))
r=l.get("roofline",{})
print("$wl", l["value"], l["unit"], "| roofline", r.get("achieved"), r.get("unit"), "frac", r.get("frac"), "| kernel us", r.get("avg_launch_us"))
PY
done
