#!/bin/bash
# resident bi-LSTM kernel: S2S parity tests, then the cfg-5 bench with and without it
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x -k "s2s or seq2seq or cfg5 or Seq2Seq" 2>&1 | tail -15
for v in 1 0 1 0; do
  echo "MMK_S2S_SEQ=$v"
  timeout 300 python bench.py --tuning MMK_S2S_SEQ=$v --workload s2s_cfg5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['us_per_generate_step'], d['roofline']['frac'])"
done
