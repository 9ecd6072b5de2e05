#!/bin/bash
# Measurement builds of libairfe.so (never shipped; *.so.tmp is git-ignored but travels with gpurun):
#   airslam_amd/libairfe_T.so.tmp    -DLF_TIMING    per-phase timers of the LightGlue block   (tools/lf_timing.py)
#   airslam_amd/libairfe_T2.so.tmp   -DLF_TIMING2   inside the K loop of the block's ffn.0    (tools/lf_timing2.py)
#   airslam_amd/libairfe_AT.so.tmp   -DATT_TIMING   per-phase timers of attention32_kernel    (tools/att_timing.py)
#   airslam_amd/libairfe_F0.so.tmp   -DLF_FRAG=0    the block kernel on the slab image (A/B against the fragment-order weights: tools/gpu_ab_lib.sh)
set -e
cd "$(dirname "$0")/.."
python -m airslam_amd.build > /dev/null
B=airslam_amd/csrc/build
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
others() { ls $B/*.o | grep -v "/$1.o"; }
hipcc $FL -DLF_TIMING -c airslam_amd/csrc/kernels_lgblockf.hip -o /tmp/lf_T.o
hipcc --offload-arch=gfx950 -shared -fPIC -o airslam_amd/libairfe_T.so.tmp /tmp/lf_T.o $(others kernels_lgblockf)
hipcc $FL -DLF_TIMING2 -c airslam_amd/csrc/kernels_lgblockf.hip -o /tmp/lf_T2.o
hipcc --offload-arch=gfx950 -shared -fPIC -o airslam_amd/libairfe_T2.so.tmp /tmp/lf_T2.o $(others kernels_lgblockf)
hipcc $FL -DATT_TIMING -c airslam_amd/csrc/kernels_attn.hip -o /tmp/att_T.o
hipcc --offload-arch=gfx950 -shared -fPIC -o airslam_amd/libairfe_AT.so.tmp /tmp/att_T.o $(others kernels_attn)
# A/B build: the block kernel reading the slab image (round 5's layout) instead of the fragment-order copy
hipcc $FL -DLF_FRAG=0 -c airslam_amd/csrc/kernels_lgblockf.hip -o /tmp/lf_F0.o
hipcc --offload-arch=gfx950 -shared -fPIC -o airslam_amd/libairfe_F0.so.tmp /tmp/lf_F0.o $(others kernels_lgblockf)
# diagnostic: the block kernel without the weight stream of its K loops (garbage results; kernel times only)
hipcc $FL -DLF_NOWEIGHTS -c airslam_amd/csrc/kernels_lgblockf.hip -o /tmp/lf_NW.o
hipcc --offload-arch=gfx950 -shared -fPIC -o airslam_amd/libairfe_NW.so.tmp /tmp/lf_NW.o $(others kernels_lgblockf)
# diagnostic: attention without its K / V staging behind the first tile (garbage results; kernel times only)
hipcc $FL -DATT_NODMA -c airslam_amd/csrc/kernels_attn.hip -o /tmp/att_ND.o
hipcc --offload-arch=gfx950 -shared -fPIC -o airslam_amd/libairfe_ND.so.tmp /tmp/att_ND.o $(others kernels_attn)
# diagnostic: the block kernel without the B-fragment reads of its K loops (NL), and without both streams (NB): garbage results; kernel times only
hipcc $FL -DLF_NOLDS -c airslam_amd/csrc/kernels_lgblockf.hip -o /tmp/lf_NL.o
hipcc --offload-arch=gfx950 -shared -fPIC -o airslam_amd/libairfe_NL.so.tmp /tmp/lf_NL.o $(others kernels_lgblockf)
hipcc $FL -DLF_NOLDS -DLF_NOWEIGHTS -c airslam_amd/csrc/kernels_lgblockf.hip -o /tmp/lf_NB.o
hipcc --offload-arch=gfx950 -shared -fPIC -o airslam_amd/libairfe_NB.so.tmp /tmp/lf_NB.o $(others kernels_lgblockf)
# attention: the sub-tile pipeline (S1: a tested alternative, the same bits), three more deletions (garbage results; kernel times only): no staging + no barrier, no exponentials, all three;
# workgroups that do nothing (EM) / only their prologue (PO); the prologue in rounds 3-5's order (QF)
for v in "S1:-DATT_SUBTILE=1" "ANB:-DATT_NODMA -DATT_NOBAR" "ANE:-DATT_NOEXP" "ANA:-DATT_NODMA -DATT_NOBAR -DATT_NOEXP" "EM:-DATT_EMPTY" "PO:-DATT_PROLOGUE_ONLY" "QF:-DATT_Q_FIRST"; do
  n=${v%%:*}; f=${v#*:}
  hipcc $FL $f -c airslam_amd/csrc/kernels_attn.hip -o /tmp/att_$n.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o airslam_amd/libairfe_$n.so.tmp /tmp/att_$n.o $(others kernels_attn)
done
ls -la airslam_amd/*.so.tmp
