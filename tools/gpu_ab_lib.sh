#!/bin/bash
# Kernel-level A B A B of a variant library against the shipped one on ONE box: rocprofv3 kernel durations of the point-only step.
#   tools/gpu_ab_lib.sh <tag> <name> <variant .so.tmp> [bench args]      -> gpurun_out/<tag>/ab_<name>.txt
cd "$(dirname "$0")/.."
TAG=$1; NAME=$2; VAR=$3; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT
cp airslam_amd/libairfe.so /tmp/ab_main.so
{ for round in 1 2; do
    cp /tmp/ab_main.so airslam_amd/libairfe.so
    KFILTER=${KFILTER:-lg_blockf,attention32} tools/gpu_visit.sh $TAG kstats ${NAME}_main_$round --detector superpoint "$@"
    cp $VAR airslam_amd/libairfe.so
    KFILTER=${KFILTER:-lg_blockf,attention32} tools/gpu_visit.sh $TAG kstats ${NAME}_variant_$round --detector superpoint "$@"
  done; } 2>&1 | tee $OUT/ab_$NAME.txt
cp /tmp/ab_main.so airslam_amd/libairfe.so
