#!/bin/bash
# Round 5: the remaining bench lines on the final tree (the same commands as tools/gpu_r5_final.sh: sequence sweep, track x2, frontend, SuperGlue).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05h; mkdir -p $OUT
timeout 600 python bench.py --workload seq --sweep --sequences 8 > $OUT/bench_seq.json 2> $OUT/bench_seq.err; echo "seq rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_seq.json"))
for S, r in d["sweep"].items():
    print("  seq S=%s: %.1f frames/s, p50 %.3f p99 %.3f ms" % (S, r["frames_per_s"], r["latency_ms"]["p50"], r["latency_ms"]["p99"]))
print("  cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["same_schedule_as_gpu"])
PY
timeout 300 python bench.py --workload track > $OUT/bench_track.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/bench_track.json')); print('track:', round(d['value'],1), d['unit'], d['config']['detector'])"
timeout 300 python bench.py --workload track --detector plnet > $OUT/bench_track_plnet.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/bench_track_plnet.json')); print('track plnet:', round(d['value'],1), d['unit'])"
timeout 300 python bench.py --workload frontend > $OUT/bench_frontend.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/bench_frontend.json')); print('frontend:', round(d['value'],1), d['unit'])"
timeout 300 python bench.py --matcher superglue > $OUT/bench_sg.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/bench_sg.json')); print('superglue:', round(d['value'],1), round(d['ms_per_step'],3))"
timeout 300 python bench.py --detector superpoint > $OUT/bench_points_only.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/bench_points_only.json')); print('points only:', round(d['value'],1), round(d['ms_per_step'],3))"
