#!/bin/bash
# the whole GPU suite on the tree as it is, smoke(), and three default runs of the Seq2Seq bench (gpurun_out/r03/bench_s2s_cfg5_<i>.json)
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
timeout 2300 python -m pytest tests -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for i in 1 2 3; do
  timeout 300 python bench.py --workload s2s_cfg5 > gpurun_out/r03/bench_s2s_cfg5_$i.json 2>/dev/null
  python -c "import json; d=json.load(open('gpurun_out/r03/bench_s2s_cfg5_$i.json')); print('s2s run $i', d['value'], d['ms_per_step'], d['roofline']['us_per_generate_step'], d['roofline']['frac'], d['cpu_baseline'].get('matches_gpu_output'))"
done
