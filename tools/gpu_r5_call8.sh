#!/bin/bash
# Round 5: stage 1 on fp16 (hi, lo) pairs, second cut (2-way head on the accumulators, junction projections on the same pipe): parity, A B A B, phase timing, kernel trace.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05n; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_stage1_split.py tests/test_gpu_plnet_batch.py tests/test_gpu_lines.py tests/test_gpu_ref_pin.py tests/test_gpu_plnet_s0.py -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log | cut -c1-400
for lp in 3 3; do
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
grep -i "plnet_s1\|s1_junc\|s1h_junc\|wireframe\|line_filter\|s1_" $OUT/kernel_stats_lp3.csv | cut -c1-200
cp airslam_amd/libairfe.so /tmp/libairfe.keep
timeout 300 python tools/s1_timing.py 3 > $OUT/s1_timing_lp3.txt 2>&1; cat $OUT/s1_timing_lp3.txt | tail -12
cp /tmp/libairfe.keep airslam_amd/libairfe.so
