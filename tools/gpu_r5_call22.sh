#!/bin/bash
# Round 5: kernel durations of the POINT-ONLY step (no line path beside the matcher: every kernel's span is its own) on the final tree, for the kernel table of DESIGN.md.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05z; mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --detector superpoint --steps 8 --warmup 2 --cpu-pairs 0 --no-profile --stage-steps 0 > $OUT/bench_under_rocprof.json 2> $OUT/err.txt
python tools/rocpd_summary.py /tmp/kt/kt_results.db $OUT/points_only_kernel_stats.csv
head -30 $OUT/points_only_kernel_stats.csv | cut -c1-160
