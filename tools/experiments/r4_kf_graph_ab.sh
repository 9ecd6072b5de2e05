mkdir -p gpurun_out/r04kf
timeout 300 python -m pytest tests/test_gpu_keyframe.py -x -q -m gpu 2>&1 | tail -15
for g in 0 1; do
  timeout 300 python bench.py --workload b1 --steps 200 --warmup 20 --tuning kf_graph=$g > gpurun_out/r04kf/bench_b1_graph$g.json 2> gpurun_out/r04kf/bench_b1_graph$g.err
  python -c "
import json; d=json.loads(open('gpurun_out/r04kf/bench_b1_graph$g.json').read().strip().splitlines()[-1]); l=d['latency_ms']; print('graph $g', 'one call', l['pair'], 'two', l['two_calls']['pair']['p50'], l['two_calls']['detect_stereo']['p50'], 'three', l['three_calls']['pair']['p50'])"
done
