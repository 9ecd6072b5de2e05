# A/B of library variants (airslam_amd/libairfe_<name>.so.tmp against the built library) on one box: point-only bench with its stage table
set -u
cd $GRAFT_REPO_ROOT
cp airslam_amd/libairfe.so /tmp/main.so
for v in main "$@" main "$@"; do
  [ $v = main ] && cp /tmp/main.so airslam_amd/libairfe.so || cp airslam_amd/libairfe_$v.so.tmp airslam_amd/libairfe.so
  python bench.py --detector superpoint --steps 100 --cpu-pairs 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],3), {k: round(v['ms_per_step'],3) for k,v in d['stages'].items() if k in ('lg_gemm','lg_attention')})"
done
cp /tmp/main.so airslam_amd/libairfe.so
