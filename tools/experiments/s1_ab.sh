set -u
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_plnet_batch.py tests/test_gpu_stereo.py tests/test_gpu_plnet_s0.py -x -q 2>&1 | tail -3
for v in 0 1 0 1; do
AIRFE_OVERLAP_EARLY=$v python bench.py --steps 60 --cpu-pairs 0 --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('OVERLAP_EARLY=$v', round(d['value'],1), round(d['ms_per_step'],3), d['config']['lines_mean'], d['config']['junctions_mean_left'], d['config']['matches_mean'])"
done
