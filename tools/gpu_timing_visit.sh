#!/bin/bash
# The three measurement builds (tools/build_timing_variants.sh, built in the container) run on the box: each tool copies its variant over libairfe.so of the scratch
# copy; the plain library is put back at the end.   tools/gpu_timing_visit.sh <tag> [pairs]
cd "$(dirname "$0")/.."
TAG=${1:-r06t}; OUT=gpurun_out/$TAG; mkdir -p $OUT; P=${2:-64}
cp airslam_amd/libairfe.so /tmp/main.so
timeout 300 python tools/lf_timing.py $P > $OUT/lf_timing.txt 2>&1; echo "lf_timing rc=$?"; tail -16 $OUT/lf_timing.txt
timeout 300 python tools/lf_timing2.py $P > $OUT/lf_timing2.txt 2>&1; echo "lf_timing2 rc=$?"; tail -12 $OUT/lf_timing2.txt
timeout 300 python tools/att_timing.py $P 8 > $OUT/att_timing.txt 2>&1; echo "att_timing rc=$?"; tail -26 $OUT/att_timing.txt
cp /tmp/main.so airslam_amd/libairfe.so
