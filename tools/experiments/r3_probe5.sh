#!/bin/bash
# round 3: attention A/B on one box (library variants airslam_amd/libairfe_<name>.so.tmp against the built library): hashes must be equal
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
cp airslam_amd/libairfe.so /tmp/main.so
for v in main attring; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  echo "== hashes $v"; timeout 300 python tools/experiments/attn_hash.py 2>&1 | grep -v amdgpu.ids | tail -1
done
echo "== A/B (default bench, stage table)"
for v in main attring main attring; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  python bench.py --steps 60 --cpu-pairs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'points-only', round(d['config']['points_only_pairs_per_s'],1), {k: round(v['ms_per_step'],3) for k,v in d['stages'].items() if k in ('lg_gemm','lg_attention')})"
done
cp /tmp/main.so airslam_amd/libairfe.so
} > gpurun_out/r3_probe5.log 2>&1
cat gpurun_out/r3_probe5.log | tail -40
