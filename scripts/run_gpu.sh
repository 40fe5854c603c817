#!/bin/bash
# build in-tree (the .so travels with the snapshot), then hand the command to gpurun
set -e
cd "$(dirname "$0")/.."
python -m mimikit_amd.build > /dev/null
exec /usr/local/graft/bin/gpurun --timeout ${GPU_TIMEOUT:-1800} -- "$@"
