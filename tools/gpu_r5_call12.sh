#!/bin/bash
# Round 5: expf_like_glibc as glibc's algorithm, the descriptor head's gather in the streaming GEMM (desc_gather_stream), fold_out_proj without XOV.
# The whole -m gpu suite (three workers, by file: most of its time is the CPU oracle), then A B A B of desc_gather_stream on the 64-pair step.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05p; mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu -n 3 --dist loadfile > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log | cut -c1-400
for t in "" "desc_gather_stream=0" "" "desc_gather_stream=0"; do
  timeout 300 python bench.py --steps 40 --warmup 5 --cpu-pairs 0 ${t:+--tuning $t} > $OUT/bench.json 2> $OUT/bench.err
  python - "$t" <<PY
import json, sys
d = json.load(open("$OUT/bench.json"))
s = d["stages"]
print("[%-22s] %.1f pairs/s %.3f ms; head_gemm %.4f ms sample_desc %.4f lg_gemm %.4f; points-only %.1f; matches %.2f" % (sys.argv[1] or "default (stream)", d["value"], d["ms_per_step"], s["head_gemm"]["ms_per_step"], s["sample_desc"]["ms_per_step"], s["lg_gemm"]["ms_per_step"], d["config"]["points_only_pairs_per_s"], d["config"]["matches_mean"]))
PY
done 2>&1 | tee $OUT/desc_gather_ab.txt
