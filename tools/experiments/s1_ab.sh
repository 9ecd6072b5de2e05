set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_plnet_s0.py tests/test_gpu_plnet_batch.py -x -q 2>&1 | tail -3
bash tools/experiments/variants.sh 2>&1 | tail -2
