set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_plnet_batch.py tests/test_gpu_stereo.py -x -q 2>&1 | tail -3
for v in 0 1 0 1; do
AIRFE_OVERLAP_LINES=$v python bench.py --steps 60 --cpu-pairs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('OVERLAP_LINES=$v', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['frac'],3), {k: round(v['ms_per_step'],3) for k,v in d['stages'].items() if k.startswith('plnet') or k in ('lg_gemm','lg_attention')})"
done
