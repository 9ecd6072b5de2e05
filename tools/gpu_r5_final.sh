#!/bin/bash
# Round 5, final GPU visit on the round's final tree: the whole -m gpu suite from an empty diag/, smoke(), the rocprofv3 passes (kernel trace + stats, PMC, HBM traffic —
# each --pmc pass in its own run with --kernel-trace only), then the bench lines; the counter summaries are copied into profiles/ ON THE BOX first, so that the default
# line carries counters taken on the very sources it runs (roofline.counters_age).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05f; mkdir -p $OUT; rm -rf gpurun_out/diag
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
bash tools/gpu_profile.sh r05 > $OUT/profile.log 2>&1; echo "profile rc=$?"; tail -12 $OUT/profile.log | cut -c1-220
cp gpurun_out/prof_r05/pmc_summary.json profiles/r05_pmc_summary.json; cp gpurun_out/prof_r05/hbm_traffic.json profiles/r05_hbm_traffic.json
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
r = d["roofline"]
print("default:", round(d["value"], 1), "pairs/s", round(d["ms_per_step"], 3), "ms; roofline", round(r["frac"], 3), "step_frac", round(r["step_frac"], 3), "traffic", r["traffic"], "util", r["mfma_util_counters"] and round(r["mfma_util_counters"]["encoder_time_weighted"], 3), "cpu", round(d["cpu_baseline"]["value"], 2), d["cpu_baseline"].get("parity_ok"))
print({k: round(v["ms_per_step"], 3) for k, v in d["stages"].items()})
print(r["counters_age"])
PY
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/bench_driver_form.json')); print('driver form (--steps 20 --warmup 5):', round(d['value'],1), round(d['ms_per_step'],3))"
timeout 600 python bench.py --workload seq --sweep --sequences 8 > $OUT/bench_seq.json 2> $OUT/bench_seq.err; echo "seq rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_seq.json"))
for S, r in d["sweep"].items():
    print("  seq S=%s: %.1f frames/s, p50 %.3f p99 %.3f ms" % (S, r["frames_per_s"], r["latency_ms"]["p50"], r["latency_ms"]["p99"]), r["wall_split_ms_per_step"])
print("  cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["same_schedule_as_gpu"], d["config"]["schedule"])
PY
timeout 300 python bench.py --workload track > $OUT/bench_track.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/bench_track.json')); print('track:', round(d['value'],1), d['unit'], d['config']['detector'])"
timeout 300 python bench.py --workload track --detector plnet > $OUT/bench_track_plnet.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/bench_track_plnet.json')); print('track plnet:', round(d['value'],1), d['unit'])"
timeout 300 python bench.py --workload frontend > $OUT/bench_frontend.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/bench_frontend.json')); print('frontend:', round(d['value'],1), d['unit'])"
timeout 300 python bench.py --workload b1 --steps 300 --warmup 20 > $OUT/bench_b1.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/bench_b1.json')); l=d['latency_ms']; print('b1: keyframe p50 %.4f p99 %.4f; two calls %.4f; three %.4f; tracked frame %.4f; with temporal %.4f; agree %s' % (l['pair']['p50'], l['pair']['p99'], l['two_calls']['pair']['p50'], l['three_calls']['pair']['p50'], l['tracked_frame']['one_call']['p50'], l['keyframe_with_temporal_match']['one_call']['p50'], d['call_forms_agree']))"
