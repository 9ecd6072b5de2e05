#!/bin/bash
# round 3: the line path's stream at the lowest priority against equal priorities
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
cp airslam_amd/libairfe.so /tmp/main.so
for v in main noprio main noprio main noprio; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  python bench.py --steps 100 --cpu-pairs 0 --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), 'points-only', round(d['config']['points_only_pairs_per_s'],1), d['config']['lines_mean'], d['config']['matches_mean'])"
done
cp /tmp/main.so airslam_amd/libairfe.so
} > gpurun_out/r3_probe10.log 2>&1
cat gpurun_out/r3_probe10.log
