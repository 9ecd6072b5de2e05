#!/bin/bash
# round 3: smoke(), the side workloads on the final build, a determinism soak on one more box
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
echo "== sweep"; bash tools/experiments/sweep.sh 2>&1 | grep -v amdgpu.ids
echo "== soak: trace off, 3000 steps per stream arrangement (device-side compare)"
for ov in 1 0; do AIRFE_OVERLAP_LINES=$ov timeout 600 python tools/experiments/soak.py 3000 2>&1 | grep -v amdgpu.ids | tail -1; done
} > gpurun_out/r3_probe9.log 2>&1
cat gpurun_out/r3_probe9.log
