#!/bin/bash
# Round 5: the fused block's second round in 96-token passes (every CU 13 token tiles instead of 14 / 7): bit-identity with uniform passes, then kernel durations,
# uniform (--tuning lgb_tokens=112) / default (mixed) / uniform / default.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05x; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lightglue.py -q -m gpu -k "two_round or tile_sizes or bench_size or folded or vs_oracle" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log | cut -c1-300
run() {   # $1 = label, $2 = tuning
  rm -rf /tmp/kt
  rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --detector superpoint --steps 4 --warmup 2 --cpu-pairs 0 --no-profile --stage-steps 0 ${2:+--tuning $2} > /dev/null 2> $OUT/err_$1.txt
  python tools/rocpd_summary.py /tmp/kt/kt_results.db $OUT/ks_$1.csv > /dev/null 2>&1
  python - "$OUT/ks_$1.csv" "$1" <<'PY'
import csv, sys
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "lg_blockf" in n:
        tot += float(r["TotalDurationNs"]) / 1e3
        print("  %-8s %-60s calls %4s avg %9.2f us min %9.2f max %9.2f" % (sys.argv[2], n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
print("  %-8s lg_blockf per step: %.1f us" % (sys.argv[2], tot / 6.0))
PY
}
{ run uniform lgb_tokens=112; run mixed ""; run uniform2 lgb_tokens=112; run mixed2 ""; } 2>&1 | tee $OUT/mixed_ab.txt
