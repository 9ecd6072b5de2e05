#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r06ag
for g in 1 2 3 4; do
  for q in default 2 8; do
    if [ "$q" = "default" ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    timeout 300 python bench.py --workload seq --sequences 8 --groups $g --frames 200 --cpu-pairs 0 > gpurun_out/r06ag/seq_g${g}_q${q}.json 2> gpurun_out/r06ag/seq_g${g}_q${q}.err
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r06ag/seq_g${g}_q${q}.json")); print("groups=${g} hwq=${q}: %.0f frames/s, %.3f ms per time-step" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("groups=${g} hwq=${q}: failed", e)
PY
  done
done
