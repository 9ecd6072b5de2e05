#!/bin/bash
# round 3: the line branch's 3x3 conv inside the encoder's sequence (main) against on the side stream beside the matcher (convlate)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
cp airslam_amd/libairfe.so /tmp/main.so
for v in main convlate; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  echo "== hashes $v"; timeout 300 python tools/experiments/step_hash.py 2>&1 | grep -v amdgpu.ids | tail -1
done
for v in main convlate main convlate main convlate; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  python bench.py --steps 100 --cpu-pairs 0 --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'points-only', round(d['config']['points_only_pairs_per_s'],1))"
done
cp /tmp/main.so airslam_amd/libairfe.so
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_plnet_batch.py tests/test_gpu_plnet_s0.py tests/test_gpu_bench_contract.py -q 2>&1 | tail -3
} > gpurun_out/r3_probe12.log 2>&1
cat gpurun_out/r3_probe12.log
