#!/bin/bash
# round 5: phase stamps of the resident SampleRNN kernel (diagnostic build), cfg 3
export TMPDIR=/tmp
mkdir -p gpurun_out/r05a
MMK_DIAG_LIB=1 MMK_SRNN_STAMPS=1 timeout 300 python scripts/srnn_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05a/srnn_stamps.log
