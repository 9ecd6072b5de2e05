#!/bin/bash
# Round 5: lg_blockf with two slabs of weights in flight in the 32-feature GEMMs of the large token tiles (libairfe_d1.so.tmp = -DLF_DEPTH2=1): bit-identity
# tests on the variant, then kernel durations under rocprofv3, main / d1 / main / d1.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05t; mkdir -p $OUT
export TMPDIR=/tmp
cp airslam_amd/libairfe.so /tmp/main.so
run() {   # $1 = label
  rm -rf /tmp/kt
  rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --detector superpoint --steps 4 --warmup 2 --cpu-pairs 0 --no-profile --stage-steps 0 > /dev/null 2> $OUT/err_$1.txt
  python tools/rocpd_summary.py /tmp/kt/kt_results.db $OUT/ks_$1.csv > /dev/null 2>&1
  echo "== $1"
  python - "$OUT/ks_$1.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("lg_blockf",)):
        print("  %-64s calls %4s avg %9.1f us min %9.1f max %9.1f" % (n.split("(")[0][-64:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}
cp airslam_amd/libairfe_d1.so.tmp airslam_amd/libairfe.so
timeout 600 python -m pytest tests/test_gpu_lightglue.py -q -m gpu -k "tile_sizes or folded or out_projection or vs_oracle" > $OUT/pytest.log 2>&1; echo "pytest (d1) rc=$?"; tail -3 $OUT/pytest.log | cut -c1-300
{ cp /tmp/main.so airslam_amd/libairfe.so; run main; cp airslam_amd/libairfe_d1.so.tmp airslam_amd/libairfe.so; run d1; cp /tmp/main.so airslam_amd/libairfe.so; run main2; cp airslam_amd/libairfe_d1.so.tmp airslam_amd/libairfe.so; run d1b; } 2>&1 | tee $OUT/depth2_ab.txt
cp /tmp/main.so airslam_amd/libairfe.so
