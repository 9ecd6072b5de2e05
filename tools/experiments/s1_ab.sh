set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 0 1 0 1; do
AIRFE_DEC_SMALL=$v python bench.py --detector plnet --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('DEC_SMALL=$v', round(d['value'],1), round(d['ms_per_step'],3), {k: round(v['ms_per_step'],3) for k,v in d['stages'].items() if k.startswith('plnet') or k in ('head_gemm','conv3x3_cin128')})"
done
AIRFE_DEC_SMALL=1 python -m pytest tests/test_gpu_plnet_batch.py tests/test_gpu_plnet_s0.py -x -q 2>&1 | tail -2
