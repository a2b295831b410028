#!/bin/bash
# A/B of the forward's two-entries-per-step walk (default build) against one entry per step (ilp1 variant)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
V=$ROOT/opensplat_amd/csrc/libgsplat_hip_ilp1.so
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_ilp2.log 2>&1; tail -3 $OUT/pytest_ilp2.log
L=$OUT/ab_fwd_ilp_r04.log; : > $L
for r in 1 2; do
for lib in default ilp1; do
  if [ $lib = default ]; then unset GSPLAT_HIP_LIB; else export GSPLAT_HIP_LIB=$V; fi
  for cfg in c2 c3; do
    timeout 300 python bench.py --no-cpu-baseline --config $cfg --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $cfg run$r', round(d['value'],1), '/s', {k:round(v*1e3,1) for k,v in d['kernel_ms'].items()})" >> $L
  done
done
done
unset GSPLAT_HIP_LIB
timeout 300 python scripts/timeline_sweep.py 384 288 > $OUT/timeline_sweep_384_ilp2_r04.json 2> $OUT/ts_a.err
timeout 300 python scripts/timeline_sweep.py 96 72 > $OUT/timeline_sweep_96_ilp2_r04.json 2> $OUT/ts_b.err
cat $L; grep -h gaussians $OUT/ts_a.err $OUT/ts_b.err
