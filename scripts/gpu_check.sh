#!/bin/bash
# One gpurun call: GPU parity tests + smoke + a short bench; logs land in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
