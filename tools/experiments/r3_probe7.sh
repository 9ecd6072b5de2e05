#!/bin/bash
# round 3: encoder chunks that straddle the left / right sources: 128 images per launch of the full-resolution layers against 64
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== regression"; timeout 600 python -m pytest tests/test_gpu_stereo.py tests/test_gpu_plnet_batch.py tests/test_gpu_detector.py -q 2>&1 | tail -4
for ch in 64 128 64 128; do
  python bench.py --steps 60 --cpu-pairs 0 --chunk $ch 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chunk $ch', round(d['value'],1), round(d['ms_per_step'],3), 'points-only', round(d['config']['points_only_pairs_per_s'],1), {k: round(v['ms_per_step'],3) for k,v in d['stages'].items() if k in ('preprocess','conv3x3_cin64','conv3x3_cin128')}, 'frac', round(d['roofline']['frac'],3))"
done
} > gpurun_out/r3_probe7.log 2>&1
cat gpurun_out/r3_probe7.log | tail -12
