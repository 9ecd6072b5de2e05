#!/bin/bash
# Round 5: PLNet stage 1 with fp16 (hi, lo) operand pairs on the 2-byte matrix pipe (cfg.line_precision = 3): parity tests, then A B A B against the f32-input MFMA form.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05h; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stage1_split.py -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.log | cut -c1-400
for lp in 2 3 2 3; do
  timeout 300 python bench.py --steps 60 --warmup 5 --cpu-pairs 0 --line-precision $lp > $OUT/bench_lp$lp.json 2> $OUT/bench_lp$lp.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_lp$lp.json"))
print("line_precision=$lp: %.1f pairs/s %.3f ms; plnet_stage1 %.4f ms; lines %.2f; matches %.2f" % (d["value"], d["ms_per_step"], d["stages"]["plnet_stage1"]["ms_per_step"], d["config"]["lines_mean"], d["config"]["matches_mean"]))
PY
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --steps 5 --warmup 2 --cpu-pairs 0 --no-profile --line-precision 3 > /dev/null 2> $OUT/kt.err
python tools/rocpd_summary.py $OUT/kt/kt_results.db $OUT/kernel_stats_lp3.csv > /dev/null 2>&1; rm -rf $OUT/kt
grep -i "plnet_s1\|s1_junc" $OUT/kernel_stats_lp3.csv | cut -c1-200
