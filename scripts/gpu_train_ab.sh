cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in default b; do
  if [ "$v" = default ]; then unset GSPLAT_HIP_LIB; else export GSPLAT_HIP_LIB=$PWD/opensplat_amd/csrc/libgsplat_hip_$v.so; fi
  python scripts/bench_train_step.py --ours-only 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); it=d['iteration']; print('$v', round(it['ms'],4), {k:round(v,3) for k,v in it['stage_ms'].items()})"
done
done
