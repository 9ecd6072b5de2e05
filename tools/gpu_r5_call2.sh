#!/bin/bash
# Round 5, second GPU visit: the WHOLE -m gpu suite (every failure wanted), the assignment A/B with the transposing reductions, batch-1 latency, the shared-S probe,
# a kernel trace of the default step.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05b; mkdir -p $OUT
export TMPDIR=/tmp
./tools/microbench/p_transpose > $OUT/probe_shared_s_transpose.txt 2>&1; cat $OUT/probe_shared_s_transpose.txt
timeout 1700 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/pytest.log
for f in 1 0 1 0; do
  timeout 300 python bench.py --steps 60 --warmup 5 --cpu-pairs 0 --tuning assign_fused=$f > $OUT/bench_assign_fused$f.json 2> $OUT/bench_assign_fused$f.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_assign_fused$f.json"))
print("assign_fused=$f: %.1f pairs/s %.3f ms; lg_assign %.4f ms; lg_gemm %.3f; matches %.2f" % (d["value"], d["ms_per_step"], d["stages"]["lg_assign"]["ms_per_step"], d["stages"]["lg_gemm"]["ms_per_step"], d["config"]["matches_mean"]))
PY
done
for f in 1 0; do
  timeout 300 python bench.py --workload b1 --steps 200 --warmup 20 --tuning assign_fused=$f > $OUT/bench_b1_assign_fused$f.json 2> $OUT/bench_b1_$f.err
  python -c "import json; d=json.load(open('$OUT/bench_b1_assign_fused$f.json')); l=d['latency_ms']; print('b1 assign_fused=$f: keyframe p50 %.4f p99 %.4f; tracked frame %.4f; with temporal %.4f' % (l['pair']['p50'], l['pair']['p99'], l['tracked_frame']['one_call']['p50'], l['keyframe_with_temporal_match']['one_call']['p50']))"
done
rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python bench.py --steps 5 --warmup 2 --cpu-pairs 0 --no-profile > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
python tools/rocpd_summary.py $OUT/kt/kt_results.db $OUT/kernel_stats.csv > /dev/null 2>&1; rm -rf $OUT/kt
grep -i "lg_sim\|sim_kernel\|lg_filter\|lg_row\|lg_col\|rowdot" $OUT/kernel_stats.csv | head
