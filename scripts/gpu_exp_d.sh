#!/bin/bash
# round 6, session 3: records zeroed behind gs_gaussian_backward's read instead of the per-frame memset (re-measured on
# round 6's kernels), then the extended random sweep of the compositing kernels
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06s3
B="python bench.py --no-cpu-baseline --steps 40 --warmup 5"
pr() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()}, {k:round(v,3) for k,v in d['stage_ms'].items()})"; }
for rep in 1 2 3; do
  $B 2>/dev/null | pr memset
  GSPLAT_RECORDS_ZEROED=1 $B 2>/dev/null | pr zeroed
done 2>&1 | tee gpurun_out/r06s3/ab_records_zeroed.log
SWEEP_TAG=r06 SWEEP_RANGE="15 1215" SWEEP_TIMEOUT=1500 bash scripts/gpu_sweep.sh
