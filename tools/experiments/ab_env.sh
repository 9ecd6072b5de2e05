# A/B of an environment switch on one box: bench.py (point-only and default) with the switch off / on, twice
set -u
cd $GRAFT_REPO_ROOT
VAR=$1
for v in 0 1 0 1; do
  env $VAR=$v python bench.py --detector superpoint --steps 100 --cpu-pairs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v points', round(d['value'],1), round(d['ms_per_step'],3), {k: round(v['ms_per_step'],3) for k,v in d['stages'].items() if k in ('lg_gemm','lg_attention')})"
done
