#!/bin/bash
# Round 5: fold_out_proj + the x tile landing under the attention half of ffn.0 (XOV).  The whole -m gpu suite (the first visit stopped at the SuperGlue pin, whose
# Python twin had the wrong K order), A B A B of the library with / without XOV (libairfe_noxov.so.tmp = -DLF_XOVERLAP=0), then the per-phase timers (-DLF_TIMING).
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05o; mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log | cut -c1-600
cp airslam_amd/libairfe.so /tmp/main.so
for v in main noxov main noxov; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  timeout 300 python bench.py --steps 40 --warmup 5 --cpu-pairs 0 > $OUT/bench.json 2> $OUT/bench.err
  python - "$v" <<PY
import json, sys
d = json.load(open("$OUT/bench.json"))
s = d["stages"]
print("[%-6s] %.1f pairs/s %.3f ms; lg_gemm %.4f ms (%.0f TF/s) attention %.4f assign %.4f; matches %.2f" % (sys.argv[1], d["value"], d["ms_per_step"], s["lg_gemm"]["ms_per_step"], s["lg_gemm"]["tflops"], s["lg_attention"]["ms_per_step"], s["lg_assign"]["ms_per_step"], d["config"]["matches_mean"]))
PY
done 2>&1 | tee $OUT/xov_ab.txt
cp /tmp/main.so airslam_amd/libairfe.so
timeout 300 python tools/lf_timing.py 64 2>&1 | tee $OUT/lf_timing.txt | tail -16
cp /tmp/main.so airslam_amd/libairfe.so
