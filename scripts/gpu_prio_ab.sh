#!/bin/bash
# round 6: issue priority of the backward waves (GS_BWD_PRIO) against the build without it, same box, interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06s5; mkdir -p $O
REPS="1 2 3" bash scripts/gpu_lib_ab.sh noprio > $O/ab_bwdprio_c2.log 2>&1
REPS="1 2" BENCH_EXTRA="--config c3" bash scripts/gpu_lib_ab.sh noprio > $O/ab_bwdprio_c3.log 2>&1
REPS="1 2" BENCH_EXTRA="--hot 0.02" bash scripts/gpu_lib_ab.sh noprio > $O/ab_bwdprio_hot.log 2>&1
GSPLAT_BWD_FLAGS=0x4000000 REPS="1 2" bash scripts/gpu_lib_ab.sh noprio > $O/ab_bwdprio_c2_fourgroup.log 2>&1
for rep in 1 2; do for v in default noprio; do
  if [ "$v" = default ]; then unset GSPLAT_HIP_LIB; else export GSPLAT_HIP_LIB=$PWD/opensplat_amd/csrc/libgsplat_hip_$v.so; fi
  python bench.py --no-cpu-baseline --steps 20 --warmup 3 --cameras-per-rank 2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'two in flight', round(d['value'],1), d.get('cameras_c1_c2') or d.get('serial_loop'))"
done; done > $O/ab_bwdprio_cpr2.log 2>&1
tail -n 20 $O/ab_bwdprio_*.log
