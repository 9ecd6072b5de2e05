#!/bin/bash
# round 3, probe 1: name the launch of the matcher that first leaves the majority result (beside the line path and on one stream)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== base rate, overlap=1, no trace"; AIRFE_OVERLAP_LINES=1 timeout 300 python tools/experiments/plnet_determinism.py 300 stereo 2>&1 | grep -v amdgpu.ids | tail -6
echo "== trace, overlap=1"; AIRFE_OVERLAP_LINES=1 timeout 600 python tools/experiments/matcher_trace.py 800 800 2>&1 | grep -v amdgpu.ids
echo "== trace, overlap=0"; AIRFE_OVERLAP_LINES=0 timeout 600 python tools/experiments/matcher_trace.py 3000 1500 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r3_probe1.log 2>&1
tail -60 gpurun_out/r3_probe1.log
