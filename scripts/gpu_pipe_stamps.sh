#!/bin/bash
# Per-phase wall-clock stamps of the pipelined WaveNet kernel (diagnostic build), one line per stage:owner:wave.
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/pipe_stamps.log
for ST in ${STAMP_SETS:-0:0:0 3:0:0 6:0:0 7:0:0 7:5:0}; do
  IFS=: read -r S O W <<< "$ST"
  echo "== stage $S owner $O wave ${W:-0}" >> gpurun_out/pipe_stamps.log
  MMK_DIAG_LIB=1 MMK_WN_STAMPS=1 MMK_WN_STAMP_STAGE=$S MMK_WN_STAMP_OWNER=$O MMK_WN_STAMP_WAVE=${W:-0} timeout 300 python bench.py --steps 1 --warmup 0 --seconds 0.064 --no-cpu-baseline 2>&1 | grep -E "stamps" | tail -1 >> gpurun_out/pipe_stamps.log
done
python - <<'PY'
import re
hdr = None
for l in open('gpurun_out/pipe_stamps.log').read().split('\n'):
    if l.startswith('=='):
        hdr = l
        continue
    m = re.findall(r"([a-zA-Z0-9 +/']+(?:\([^)]*\))?)=([0-9.]+)", l)
    if not m:
        continue
    d = {k.strip(): float(v) for k, v in m}
    print(hdr, '(us per visit) ' + ' | '.join(f"{k[:24]}={v * 1000 / 8192:.2f}" for k, v in d.items() if not k.startswith(('whole', 'shader'))))
PY
