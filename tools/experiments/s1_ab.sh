# stage-1 kernel on one box: parity tests, bench of the batched PLNet step, then the per-phase timers
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_plnet_superglue.py tests/test_gpu_plnet_s0.py tests/test_gpu_plnet_batch.py -x -q 2>&1 | tail -3
python bench.py --detector plnet --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],3), {k: round(v['ms_per_step'],3) for k,v in d['stages'].items() if k.startswith('plnet') or k=='head_gemm'})"
cp airslam_amd/libairfe.so /tmp/main.so
python tools/s1_timing.py 2>&1 | tail -12
cp /tmp/main.so airslam_amd/libairfe.so
