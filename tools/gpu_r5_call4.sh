#!/bin/bash
# Round 5, fourth GPU visit: the tests touched since the third one, the shared-S probe with its per-exchange check, the sequence sweep after the host-side work.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05d; mkdir -p $OUT
./tools/microbench/p_transpose > $OUT/probe_shared_s_transpose.txt 2>&1; cat $OUT/probe_shared_s_transpose.txt
timeout 900 python -m pytest tests/test_gpu_seq.py tests/test_gpu_bench_contract.py -q -m gpu -k "seq or default_workload or track" > $OUT/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -12 $OUT/pytest_a.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_lightglue.py -q -m gpu -k "bench_size or assignment" > $OUT/pytest_b.log 2>&1; echo "pytest b rc=$?"; tail -5 $OUT/pytest_b.log | cut -c1-300
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
