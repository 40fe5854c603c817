#!/bin/bash
# round 5: feature kernels - parity of the feature tests on the tree's library, then the bench lines of library variants (mimikit_amd/variants)
mkdir -p gpurun_out/r05f
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_features.py -m gpu -q --timeout=300 -x > gpurun_out/r05f/pytest_feat.log 2>&1
echo "pytest exit: $?"; grep -v "^E  \|^    \|^$" gpurun_out/r05f/pytest_feat.log | tail -5
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', d['config']['workload'][:34], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"; }
echo "tree:"
for WL in ${WORKLOADS:-mulaw stft istft gla}; do timeout 300 python bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line; done
cp mimikit_amd/libmmk_hip.so /tmp/libmmk_base.so
for v in mimikit_amd/variants/libmmk_*.so; do
  cp $v mimikit_amd/libmmk_hip.so
  echo "$(basename $v):"
  for WL in ${AB_WORKLOADS:-stft}; do MMK_DIAG_LIB= timeout 300 python bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | line; done
done 2>&1
cp /tmp/libmmk_base.so mimikit_amd/libmmk_hip.so
