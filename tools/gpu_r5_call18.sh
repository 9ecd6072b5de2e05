#!/bin/bash
# Round 5: lg_blockf's slab loop as two steps per trip with the fragment sets changing roles (no `cur = nxt` register moves); attention32_kernel in its
# straight-line form (the default now).  LightGlue / SuperGlue parity + bit-identity tests on the new library, then kernel durations, pp0 (one step per trip) / main x2.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05v; mkdir -p $OUT
export TMPDIR=/tmp
cp airslam_amd/libairfe.so /tmp/main.so
timeout 900 python -m pytest tests/test_gpu_lightglue.py tests/test_gpu_plnet_superglue.py tests/test_zz_gpu_determinism.py tests/test_gpu_stereo.py -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log | cut -c1-300
run() {   # $1 = label
  rm -rf /tmp/kt
  rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --detector superpoint --steps 4 --warmup 2 --cpu-pairs 0 --no-profile --stage-steps 0 > /dev/null 2> $OUT/err_$1.txt
  python tools/rocpd_summary.py /tmp/kt/kt_results.db $OUT/ks_$1.csv > /dev/null 2>&1
  python - "$OUT/ks_$1.csv" "$1" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "attention32" in n or "lg_blockf" in n:
        print("  %-6s %-56s calls %4s avg %9.2f us min %9.2f max %9.2f" % (sys.argv[2], n.split("(")[0][-56:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
}
for v in pp0 main pp0 main; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  run $v
done 2>&1 | tee $OUT/pingpong_ab.txt
cp /tmp/main.so airslam_amd/libairfe.so
