#!/bin/bash
# Round 5: the attention out-projection folded into ffn.0 / mlp.0 at pack time (airfe_tuning::fold_out_proj): the whole -m gpu suite on the folded default
# (incl. the two bit-identity pins against the Python fold with an identity out-projection), then A B A B of the 64-pair step and of batch 1.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05n; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log | cut -c1-600
for t in "" "fold_out_proj=0" "" "fold_out_proj=0"; do
  timeout 300 python bench.py --steps 40 --warmup 5 --cpu-pairs 0 ${t:+--tuning $t} > $OUT/bench.json 2> $OUT/bench.err
  python - "$t" <<PY
import json, sys
d = json.load(open("$OUT/bench.json"))
s = d["stages"]
print("[%-16s] %.1f pairs/s %.3f ms; lg_gemm %.4f ms (%.0f TF/s) attention %.4f assign %.4f; matches %.2f" % (sys.argv[1] or "default (fold)", d["value"], d["ms_per_step"], s["lg_gemm"]["ms_per_step"], s["lg_gemm"]["tflops"], s["lg_attention"]["ms_per_step"], s["lg_assign"]["ms_per_step"], d["config"]["matches_mean"]))
PY
done 2>&1 | tee $OUT/fold_out_ab.txt
for t in "" "fold_out_proj=0"; do
  timeout 300 python bench.py --workload b1 --steps 300 --warmup 20 ${t:+--tuning $t} > $OUT/bench_b1.json 2> /dev/null
  python - "$t" <<PY
import json, sys
d = json.load(open("$OUT/bench_b1.json")); l = d["latency_ms"]
print("[%-16s] b1: keyframe p50 %.4f p99 %.4f; tracked frame %.4f; with temporal %.4f; MatchingPoints call %.4f" % (sys.argv[1] or "default (fold)", l["pair"]["p50"], l["pair"]["p99"], l["tracked_frame"]["one_call"]["p50"], l["keyframe_with_temporal_match"]["one_call"]["p50"], l["three_calls"]["match"]["p50"]))
PY
done 2>&1 | tee -a $OUT/fold_out_ab.txt
timeout 300 python bench.py --matcher superglue --steps 20 --warmup 3 --cpu-pairs 0 > $OUT/bench_sg.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/bench_sg.json')); print('superglue:', round(d['value'],1), round(d['ms_per_step'],3))" | tee -a $OUT/fold_out_ab.txt
