#!/bin/bash
# Round 5, re-stamp after the last kernel change (lg_blockf_mixed_kernel): the whole -m gpu suite from an empty diag/, smoke(), the rocprofv3 passes of tools/gpu_profile.sh,
# the default bench line with the fresh counter summaries in place, the driver's form, batch 1.  (Sequence / track / frontend lines: tools/gpu_r5_final.sh, one commit earlier.)
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05g; mkdir -p $OUT; rm -rf gpurun_out/diag
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
bash tools/gpu_profile.sh r05 > $OUT/profile.log 2>&1; echo "profile rc=$?"
cp gpurun_out/prof_r05/pmc_summary.json profiles/r05_pmc_summary.json; cp gpurun_out/prof_r05/hbm_traffic.json profiles/r05_hbm_traffic.json
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
r = d["roofline"]
print("default:", round(d["value"], 1), "pairs/s", round(d["ms_per_step"], 3), "ms; roofline", round(r["frac"], 3), "step_frac", round(r["step_frac"], 3), "traffic", r["traffic"], "util", r["mfma_util_counters"] and round(r["mfma_util_counters"]["encoder_time_weighted"], 3), "cpu", round(d["cpu_baseline"]["value"], 2), d["cpu_baseline"].get("parity_ok"))
print({k: round(v["ms_per_step"], 3) for k, v in d["stages"].items()})
print(r["counters_age"]["traffic"]["stale"], r["counters_age"]["mfma_util"]["stale"])
PY
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/bench_driver_form.json')); print('driver form (--steps 20 --warmup 5):', round(d['value'],1), round(d['ms_per_step'],3))"
timeout 300 python bench.py --workload b1 --steps 300 --warmup 20 > $OUT/bench_b1.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/bench_b1.json')); l=d['latency_ms']; print('b1: keyframe p50 %.4f p99 %.4f; tracked frame %.4f; agree %s' % (l['pair']['p50'], l['pair']['p99'], l['tracked_frame']['one_call']['p50'], d['call_forms_agree']))"
