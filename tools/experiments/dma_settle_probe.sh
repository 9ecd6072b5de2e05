#!/bin/bash
# Hypothesis test for the matcher's schedule-dependent fault (lead 00): does giving the LDS-DMA tiles time to settle
# behind `s_waitcnt vmcnt(..)` (an s_sleep before the barrier) in the attention / GEMM / block kernels remove the irreproducible scores that
# appear when the line path's LDS-heavy kernels run beside the matcher?
#   step 1 (anywhere, no GPU): bash tools/experiments/dma_settle_probe.sh build    -> airslam_amd/libairfe_slp.so.tmp
#   step 2 (on the MI355X)   : bash tools/experiments/dma_settle_probe.sh run      -> runs outside the majority result, variant vs normal build
set -eu
cd "$(dirname "$0")/../.."
if [ "${1:-build}" = build ]; then
  ( cd airslam_amd/csrc
    python3 - <<'PY'
subs = {
 "kernels_attn":     [('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");', 'asm volatile("s_waitcnt vmcnt(0)\\n\\ts_sleep 8" ::: "memory");')],
 "kernels_gemm8":    [('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");', 'asm volatile("s_waitcnt vmcnt(0)\\n\\ts_sleep 8" ::: "memory");')],
 "kernels_lgblockf": [('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");', 'asm volatile("s_waitcnt vmcnt(0)\\n\\ts_sleep 8" ::: "memory");')],
 "kernels_gemmr":    [('asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");', 'asm volatile("s_waitcnt vmcnt(%0)\\n\\ts_sleep 8" ::"n"(N) : "memory");'),
                      ('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");', 'asm volatile("s_waitcnt vmcnt(0)\\n\\ts_sleep 8" ::: "memory");')],
}
for f, ss in subs.items():
    s = open(f + ".hip").read(); n = 0
    for a, b in ss:
        n += s.count(a); s = s.replace(a, b)
    assert n > 0, f
    open("_slp_" + f + ".hip", "w").write(s); print(f, n, "waits patched")
PY
    objs=""
    for f in kernels_attn kernels_gemm8 kernels_gemmr kernels_lgblockf; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c _slp_$f.hip -o /tmp/slp_$f.o; rm -f _slp_$f.hip; objs="$objs /tmp/slp_$f.o"
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libairfe_slp.so.tmp \
      $(ls build/*.o | grep -v "/kernels_attn.o\|/kernels_gemm8.o\|/kernels_gemmr.o\|/kernels_lgblockf.o") $objs )
  ls -la airslam_amd/libairfe_slp.so.tmp
else
  export AIRFE_OVERLAP_LINES=1
  cp airslam_amd/libairfe.so /tmp/airfe_main.so
  cp airslam_amd/libairfe_slp.so.tmp airslam_amd/libairfe.so
  echo "== with s_sleep behind every DMA wait"; python tools/experiments/plnet_determinism.py 200 stereo 2>&1 | grep -v amdgpu.ids | tail -2
  cp /tmp/airfe_main.so airslam_amd/libairfe.so
  echo "== normal build"; python tools/experiments/plnet_determinism.py 200 stereo 2>&1 | grep -v amdgpu.ids | tail -2
fi
