# the other bench configurations on one box, one JSON line each into gpurun_out/sweep/ (BASELINE.json configs[2], configs[4] and the side workloads)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/sweep
run() { name=$1; shift; python bench.py "$@" > gpurun_out/sweep/$name.json 2> gpurun_out/sweep/$name.err; python -c "import json; d=json.load(open('gpurun_out/sweep/$name.json')); print('$name', round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms', d['config'].get('matches_mean'), (d.get('roofline') or {}).get('frac'))"; }
run cfg2_640x480_bf16 --width 640 --height 480 --dtype bf16 --matcher-dtype bf16 --cpu-pairs 0
run sg --matcher superglue --cpu-pairs 0
run sg5 --matcher superglue --width 1280 --height 720 --max-keypoints 1024 --pairs 16 --cpu-pairs 0
run loop --workload loop --cpu-pairs 0
run points --detector superpoint --cpu-pairs 0
run pairs16 --pairs 16 --cpu-pairs 0
run pairs4 --pairs 4 --cpu-pairs 0
run plhost --plnet-host --pairs 8 --cpu-pairs 0
python tools/latency_b1.py > gpurun_out/sweep/latency_b1.json 2> gpurun_out/sweep/latency_b1.err; tail -c 300 gpurun_out/sweep/latency_b1.json; echo
