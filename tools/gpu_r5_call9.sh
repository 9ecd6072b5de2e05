#!/bin/bash
# Round 5: the kept-proposal counts taken in the j2l kernel, the junction map's 3x3 suppression inside the candidate compaction: line / junction parity, then the stage table.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05m; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_stage1_split.py tests/test_gpu_plnet_batch.py tests/test_gpu_lines.py tests/test_gpu_ref_pin.py tests/test_gpu_plnet_s0.py tests/test_gpu_keyframe.py tests/test_gpu_seq.py -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log | cut -c1-400
for i in 1 2; do
  timeout 300 python bench.py --steps 40 --warmup 5 --cpu-pairs 0 > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  python - <<PY
import json
d = json.load(open("$OUT/bench_$i.json"))
s = d["stages"]
print("%.1f pairs/s %.3f ms; plnet_s0_decode %.4f ms; plnet_stage1 %.4f ms; plnet_filter %.4f; lines %.2f; matches %.2f" % (d["value"], d["ms_per_step"], s["plnet_s0_decode"]["ms_per_step"], s["plnet_stage1"]["ms_per_step"], s["plnet_filter"]["ms_per_step"], d["config"]["lines_mean"], d["config"]["matches_mean"]))
PY
done
timeout 300 python bench.py --workload b1 --steps 300 --warmup 20 > $OUT/bench_b1.json 2> /dev/null; python -c "import json; d=json.load(open('$OUT/bench_b1.json')); l=d['latency_ms']; print('b1: keyframe p50 %.4f p99 %.4f; two calls %.4f; three %.4f; tracked frame %.4f; with temporal %.4f; agree %s' % (l['pair']['p50'], l['pair']['p99'], l['two_calls']['pair']['p50'], l['three_calls']['pair']['p50'], l['tracked_frame']['one_call']['p50'], l['keyframe_with_temporal_match']['one_call']['p50'], d['call_forms_agree']))"
