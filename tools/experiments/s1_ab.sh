set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_plnet_s0.py tests/test_gpu_plnet_batch.py tests/test_gpu_plnet_superglue.py tests/test_gpu_shim.py -x -q 2>&1 | tail -3
for v in 0 1 0 1; do
AIRFE_FUSE_DEC=$v python bench.py --steps 40 --cpu-pairs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('FUSE_DEC=$v', round(d['value'],1), round(d['ms_per_step'],3), {k: round(v['ms_per_step'],3) for k,v in d['stages'].items() if k.startswith('plnet') or k in ('head_gemm',)})"
done
