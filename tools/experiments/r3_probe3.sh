#!/bin/bash
# round 3, GPU visit 4: new parity tests, the 200-frame fp32 sequence, conv64r per-wave timers, A/B of issue-priority variants
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
echo "== new parity tests"; timeout 600 python -m pytest tests/test_gpu_plnet_s0.py tests/test_gpu_fp32.py -q 2>&1 | tail -15
echo "== fp32 sequence (BASELINE configs[1])"; timeout 900 python tools/seq_fp32_parity.py --frames 200 --cache tools/_cache/seq_oracle_200.npz --out gpurun_out/r03_seq_fp32_parity.json 2>&1 | grep -v amdgpu.ids | tail -5
echo "== A/B: issue priority variants (point-only bench, stage table)"
cp airslam_amd/libairfe.so /tmp/main.so
for v in main prio1 prio2 prio3 attprio main prio1 prio2 prio3 attprio; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  python bench.py --detector superpoint --steps 60 --cpu-pairs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), {k: round(v['ms_per_step'],3) for k,v in d['stages'].items() if k in ('conv3x3_cin64','conv3x3_cin128','lg_gemm','lg_attention')}, 'frac', round(d['roofline']['frac'],3))"
done
cp /tmp/main.so airslam_amd/libairfe.so
echo "== conv64r per-wave timers"; timeout 300 python tools/conv64r_timing.py 2>&1 | grep -v amdgpu.ids | tail -12
cp /tmp/main.so airslam_amd/libairfe.so
} > gpurun_out/r3_probe3.log 2>&1
cat gpurun_out/r3_probe3.log | tail -60
