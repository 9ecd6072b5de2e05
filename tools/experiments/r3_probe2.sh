#!/bin/bash
# round 3, probe 2: (a) what the packed-math rotation gets wrong (the round-2 build of rotate_pairs, kept as libairfe_pk.so.tmp);
# (b) the fixed build: trace + plain determinism runs, beside the line path and on one stream
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
cp airslam_amd/libairfe.so /tmp/airfe_main.so
if [ -f airslam_amd/libairfe_pk.so.tmp ]; then
  cp airslam_amd/libairfe_pk.so.tmp airslam_amd/libairfe.so
  echo "== packed-math build: trace, overlap=1"; AIRFE_OVERLAP_LINES=1 timeout 600 python tools/experiments/matcher_trace.py 300 400 2>&1 | grep -v amdgpu.ids
  cp /tmp/airfe_main.so airslam_amd/libairfe.so
fi
echo "== fixed build: trace, overlap=1"; AIRFE_OVERLAP_LINES=1 timeout 600 python tools/experiments/matcher_trace.py 1500 0 2>&1 | grep -v amdgpu.ids
echo "== fixed build: trace, overlap=0"; AIRFE_OVERLAP_LINES=0 timeout 600 python tools/experiments/matcher_trace.py 3000 0 2>&1 | grep -v amdgpu.ids
echo "== fixed build: plain determinism, overlap=1"; AIRFE_OVERLAP_LINES=1 timeout 600 python tools/experiments/plnet_determinism.py 2000 stereo 2>&1 | grep -v amdgpu.ids | tail -4
echo "== fixed build: plain determinism, overlap=0"; AIRFE_OVERLAP_LINES=0 timeout 600 python tools/experiments/plnet_determinism.py 2000 stereo 2>&1 | grep -v amdgpu.ids | tail -4
} > gpurun_out/r3_probe2.log 2>&1
tail -80 gpurun_out/r3_probe2.log
