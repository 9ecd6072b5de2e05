#!/bin/bash
# round 3: the detector head (convPb + soft-max + depth-to-space) as a streaming kernel against the tiled GEMM with the same epilogue
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
cp airslam_amd/libairfe.so /tmp/main.so
for v in main headold; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  echo "== hashes $v"; timeout 300 python tools/experiments/attn_hash.py 2>&1 | grep -v amdgpu.ids | tail -1 | sed 's/.*stereo16/stereo16/'
done
cp /tmp/main.so airslam_amd/libairfe.so
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_detector.py tests/test_gpu_stereo.py tests/test_gpu_plnet_batch.py tests/test_gpu_fp32.py -q 2>&1 | tail -3
for v in main headold main headold; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  python bench.py --steps 60 --cpu-pairs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'points-only', round(d['config']['points_only_pairs_per_s'],1), {k: round(v['ms_per_step'],3) for k,v in d['stages'].items() if k in ('head_gemm','simple_nms','conv3x3_cin128')})"
done
cp /tmp/main.so airslam_amd/libairfe.so
} > gpurun_out/r3_probe11.log 2>&1
cat gpurun_out/r3_probe11.log
