#!/bin/bash
# One GPU visit: the whole -m gpu suite (not -x: every failure is wanted), the default bench line and the track workload.
#   tools/gpu_round.sh <tag>      -> gpurun_out/<tag>_{pytest.log,bench_default.json,bench_track.json}
cd "$(dirname "$0")/.."
TAG=${1:-r03}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu ${PYTEST_ARGS:-} > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_default.json"))
print("default:", round(d["value"],1), "pairs/s", round(d["ms_per_step"],3), "ms; points-only", round(d["config"]["points_only_pairs_per_s"],1), "roofline", round(d["roofline"]["frac"],3), "cpu", round(d["cpu_baseline"]["value"],2))
print({k: round(v["ms_per_step"],3) for k,v in d["stages"].items()})
PY
timeout 600 python bench.py --workload track > gpurun_out/${TAG}_bench_track.json 2> gpurun_out/${TAG}_bench_track.err; echo "track rc=$?"
python -c "import json; d=json.load(open('gpurun_out/${TAG}_bench_track.json')); print('track:', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms', d['config'].get('matches_mean'), d['config'].get('lines_mean'))"
