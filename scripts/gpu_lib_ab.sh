#!/bin/bash
# same-box A/B of library variants, interleaved:  gpu_lib_ab.sh suffix ...  (default = in-tree library)
cd ${GRAFT_REPO_ROOT:-/root/repo}
B="python bench.py --no-cpu-baseline --steps 40 --warmup 5"
for rep in ${REPS:-1 2 3}; do
for v in default "$@"; do
  if [ "$v" = default ]; then unset GSPLAT_HIP_LIB; else export GSPLAT_HIP_LIB=$PWD/opensplat_amd/csrc/libgsplat_hip_$v.so; fi
  $B ${BENCH_EXTRA:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()}, {k:round(v,3) for k,v in d['stage_ms'].items()})"
done
done
