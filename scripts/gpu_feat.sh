#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_features.py -m gpu -q --timeout=300 -x > gpurun_out/pytest_feat.log 2>&1
echo "pytest exit: $?"; grep -v "^E  \|^    \|^$" gpurun_out/pytest_feat.log | tail -8
for WL in mulaw stft; do
  timeout 300 python bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done
