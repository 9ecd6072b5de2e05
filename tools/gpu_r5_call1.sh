#!/bin/bash
# Round 5, first GPU visit: the whole -m gpu suite on the new tree, the default bench line, the sequence workload (configs[3]) with its S sweep,
# the assignment A/B on ONE box, the runtime's last-error semantics.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05a; mkdir -p $OUT
./tools/microbench/last_error_semantics > $OUT/last_error_semantics.txt 2>&1; cat $OUT/last_error_semantics.txt
timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_zz_gpu_determinism.py > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.log
for f in 1 0; do
  timeout 300 python bench.py --steps 40 --warmup 5 --cpu-pairs 0 --tuning assign_fused=$f > $OUT/bench_assign_fused$f.json 2> $OUT/bench_assign_fused$f.err; echo "assign_fused=$f rc=$?"
done
python - <<PY
import json
for f in (1, 0):
    try:
        d = json.load(open("$OUT/bench_assign_fused%d.json" % f))
        print("assign_fused=%d: %.1f pairs/s %.3f ms; lg_assign %.4f ms; matches %.2f" % (f, d["value"], d["ms_per_step"], d["stages"]["lg_assign"]["ms_per_step"], d["config"]["matches_mean"]))
    except Exception as e:
        print("assign_fused=%d: no line (%s)" % (f, e))
PY
timeout 600 python bench.py --workload seq --sweep --sequences 8 > $OUT/bench_seq.json 2> $OUT/bench_seq.err; echo "seq rc=$?"; tail -3 $OUT/bench_seq.err
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_seq.json"))
    print("seq S=8:", round(d["value"], 1), d["unit"], d["latency_ms_per_time_step"], d["config"]["schedule"])
    for S, r in d["sweep"].items():
        print("  S=%s: %.1f frames/s, %.3f ms per time-step, p50 %.3f p99 %.3f" % (S, r["frames_per_s"], r["ms_per_time_step"], r["latency_ms"]["p50"], r["latency_ms"]["p99"]))
    print("  cpu:", d["cpu_baseline"] and (d["cpu_baseline"]["value"], d["cpu_baseline"]["same_schedule_as_gpu"]))
except Exception as e:
    print("seq: no line", e)
PY
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("default:", round(d["value"], 1), "pairs/s", round(d["ms_per_step"], 3), "ms; roofline", round(d["roofline"]["frac"], 3), "step_frac", round(d["roofline"]["step_frac"], 3), "cpu", round(d["cpu_baseline"]["value"], 2))
print({k: round(v["ms_per_step"], 3) for k, v in d["stages"].items()})
PY
