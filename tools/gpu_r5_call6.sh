#!/bin/bash
# Round 5: the two-rank launch with the side-stream gather (gloo on one GPU), the sequence sweep up to S = 32.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05g; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bench_contract.py tests/test_gpu_rccl.py -q -m gpu -k "gpus_2 or rccl or nccl or default_workload" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log | cut -c1-300
timeout 900 python bench.py --workload seq --sweep --sequences 8 > $OUT/bench_seq.json 2> $OUT/bench_seq.err; echo "seq rc=$?"; tail -2 $OUT/bench_seq.err
python - <<PY
import json
d = json.load(open("$OUT/bench_seq.json"))
for S, r in d["sweep"].items():
    print("  seq S=%s: %.1f frames/s, p50 %.3f p99 %.3f ms" % (S, r["frames_per_s"], r["latency_ms"]["p50"], r["latency_ms"]["p99"]), r["wall_split_ms_per_step"])
print("  cpu:", d["cpu_baseline"]["value"], d["cpu_baseline"]["same_schedule_as_gpu"])
PY
