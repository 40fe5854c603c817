#!/bin/bash
# A/B of library variants on the batched stage pipeline: us per step at 256 and 512 clips (scripts/bpipe_check.py --only bpipe)
export TMPDIR=/tmp
cp mimikit_amd/libmmk_hip.so /tmp/libmmk_base.so
for rep in 1 2; do
for v in mimikit_amd/variants/libmmk_*.so; do
  cp $v mimikit_amd/libmmk_hip.so
  r1=$(timeout 300 python scripts/bpipe_check.py --only bpipe --clips ${CLIPS_A:-256} --steps 256 2>/dev/null | grep -o "[0-9.]* us per step")
  r2=$(timeout 300 python scripts/bpipe_check.py --only bpipe --clips ${CLIPS_B:-512} --steps 256 2>/dev/null | grep -o "[0-9.]* us per step")
  echo "$(basename $v) ${CLIPS_A:-256}: $r1 | ${CLIPS_B:-512}: $r2"
done
done
cp /tmp/libmmk_base.so mimikit_amd/libmmk_hip.so
