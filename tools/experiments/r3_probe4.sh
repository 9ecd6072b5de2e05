#!/bin/bash
# round 3, GPU visit 5: the pipelined attention kernel — same bits as the kernel it replaces? parity suite? faster?
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
cp airslam_amd/libairfe.so /tmp/main.so
echo "== hashes, pipelined kernel"; timeout 300 python tools/experiments/attn_hash.py 2>&1 | grep -v amdgpu.ids | tail -3
cp airslam_amd/libairfe_attold.so.tmp airslam_amd/libairfe.so
echo "== hashes, round-2 kernel";   timeout 300 python tools/experiments/attn_hash.py 2>&1 | grep -v amdgpu.ids | tail -3
cp /tmp/main.so airslam_amd/libairfe.so
echo "== parity tests with the pipelined kernel"; timeout 900 python -m pytest tests/test_gpu_lightglue.py tests/test_gpu_plnet_superglue.py tests/test_gpu_stereo.py tests/test_zz_gpu_determinism.py -q -k "not keyframe" 2>&1 | tail -8
echo "== A/B (default bench, stage table)"
for v in main attold main attold; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  python bench.py --steps 60 --cpu-pairs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'points-only', round(d['config']['points_only_pairs_per_s'],1), {k: round(v['ms_per_step'],3) for k,v in d['stages'].items() if k in ('lg_gemm','lg_attention')})"
done
cp /tmp/main.so airslam_amd/libairfe.so
} > gpurun_out/r3_probe4.log 2>&1
cat gpurun_out/r3_probe4.log | tail -40
