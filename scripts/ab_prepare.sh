#!/bin/bash
# keep the current library as variant "a_before", rebuild, keep the rebuilt one as "b_after" (for scripts/gpu_ab.sh)
cd "$(dirname "$0")/.."
rm -f mimikit_amd/variants/*
cp mimikit_amd/libmmk_hip.so mimikit_amd/variants/libmmk_a_before.so
python -m mimikit_amd.build 2>&1 | tail -1
cp mimikit_amd/libmmk_hip.so mimikit_amd/variants/libmmk_b_after.so
ls mimikit_amd/variants/
