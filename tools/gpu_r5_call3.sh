#!/bin/bash
# Round 5, third GPU visit: the tests that changed since the second one, the shared-S probe, the sequence workload's sweep with its wall-time split.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05c; mkdir -p $OUT
./tools/microbench/p_transpose > $OUT/probe_shared_s_transpose.txt 2>&1; cat $OUT/probe_shared_s_transpose.txt
timeout 1200 python -m pytest tests/test_gpu_seq.py tests/test_gpu_range.py tests/test_gpu_errors.py tests/test_gpu_rccl.py tests/test_gpu_bench_contract.py tests/test_gpu_lightglue.py -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/pytest.log | cut -c1-400
timeout 600 python bench.py --workload seq --sweep --sequences 8 > $OUT/bench_seq.json 2> $OUT/bench_seq.err; echo "seq rc=$?"; tail -3 $OUT/bench_seq.err
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_seq.json"))
    print("seq S=8:", round(d["value"], 1), d["unit"], d["latency_ms_per_time_step"], d["config"]["schedule"])
    for S, r in d["sweep"].items():
        print("  S=%s: %.1f frames/s, %.3f ms per time-step, p50 %.3f p99 %.3f" % (S, r["frames_per_s"], r["ms_per_time_step"], r["latency_ms"]["p50"], r["latency_ms"]["p99"]), r["wall_split_ms_per_step"])
    print("  cpu:", d["cpu_baseline"] and (d["cpu_baseline"]["value"], d["cpu_baseline"]["same_schedule_as_gpu"]))
except Exception as e:
    print("seq: no line", e)
PY
