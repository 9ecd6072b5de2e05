#!/bin/bash
# Round 5: GELU four pairs at a time, one Horner level at a time (libairfe_g0.so.tmp = pair by pair, the form of rounds 1-4): the same bits (md5 of the
# log-assignment matrices and batched match lists under both libraries), parity tests on the new one, kernel durations g0 / main / g0 / main.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05w; mkdir -p $OUT
export TMPDIR=/tmp
cp airslam_amd/libairfe.so /tmp/main.so
{ echo -n "main: "; python tools/lib_scores_hash.py 2>/dev/null | tail -1; cp airslam_amd/libairfe_g0.so.tmp airslam_amd/libairfe.so; echo -n "g0:   "; python tools/lib_scores_hash.py 2>/dev/null | tail -1; cp /tmp/main.so airslam_amd/libairfe.so; } | tee $OUT/hash.txt
timeout 900 python -m pytest tests/test_gpu_lightglue.py -q -m gpu -k "vs_oracle or tile_sizes or folded or out_projection" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log | cut -c1-300
run() {   # $1 = label
  rm -rf /tmp/kt
  rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --detector superpoint --steps 4 --warmup 2 --cpu-pairs 0 --no-profile --stage-steps 0 > /dev/null 2> $OUT/err_$1.txt
  python tools/rocpd_summary.py /tmp/kt/kt_results.db $OUT/ks_$1.csv > /dev/null 2>&1
  python - "$OUT/ks_$1.csv" "$1" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "lg_blockf" in n:
        print("  %-6s %-56s calls %4s avg %9.2f us min %9.2f max %9.2f" % (sys.argv[2], n.split("(")[0][-56:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}
for v in g0 main g0 main; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  run $v
done 2>&1 | tee $OUT/gelu_ab.txt
cp /tmp/main.so airslam_amd/libairfe.so
