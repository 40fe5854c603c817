#!/bin/bash
# round 4: A/B of SampleRNN tier-kernel variants on the cfg-3 bench line (pass time), parity of the SampleRNN tests on the variant in $PARITY
export TMPDIR=/tmp
mkdir -p gpurun_out/r04d
cp mimikit_amd/libmmk_hip.so /tmp/libmmk_base.so
for rep in 1 2; do
for v in mimikit_amd/variants/libmmk_*.so; do
  cp $v mimikit_amd/libmmk_hip.so
  r=$(timeout 300 python bench.py --workload srnn_cfg3 --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"us_per_step": [0-9.]*' | tr '\n' ' ')
  echo "$(basename $v) $r"
done
done 2>&1 | tee gpurun_out/r04d/ab_srnn.log
if [ -n "$PARITY" ]; then
cp mimikit_amd/variants/libmmk_$PARITY.so mimikit_amd/libmmk_hip.so
timeout 1200 python -m pytest tests/test_gpu_networks.py tests/test_gpu_baseline_configs.py -q -x -k "sample_rnn or srnn or cfg1 or cfg3" 2>&1 | tail -3
fi
cp /tmp/libmmk_base.so mimikit_amd/libmmk_hip.so
