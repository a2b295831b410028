#!/bin/bash
# Same-box A/B of kernel variants selected by environment (GSPLAT_BWD_FLAGS / GSPLAT_FWD_FLAGS / GSPLAT_HIP_LIB ...):
#   gpu_flags_ab.sh TAG "VAR=val ..." "VAR=val ..."      ("" = the defaults, always run first)
# per setting: the compositing parity tests (PYTEST=full: the whole -m gpu suite), bench lines at C2 / C3 / hot spot,
# the work counters of the instrumented build (STATS=1).  ab_raster.py runs the flag sets of AB_FLAGS in one process.
set -u
TAG=${1:-f}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
SUBSET="tests/test_gpu_parity.py tests/test_gpu_deterministic.py tests/test_gpu_baseline_parity.py tests/test_gpu_ops_and_edges.py tests/test_gpu_visibility_threshold.py tests/test_gpu_parity_r03.py tests/test_gpu_c3_golden.py tests/test_gpu_fused.py"
B="python bench.py --no-cpu-baseline --steps 40 --warmup 5"
i=0
for v in "" "$@"; do
  if [ "${PYTEST:-subset}" = full ]; then
    env $v timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_${TAG}_$i.log 2>&1
  elif [ "${PYTEST:-subset}" = subset ]; then
    env $v timeout 900 python -m pytest $SUBSET -m gpu -x -q > $OUT/pytest_${TAG}_$i.log 2>&1
  fi
  [ "${PYTEST:-subset}" != none ] && echo "[$v] pytest rc=$? $(grep -E 'passed|failed|error' $OUT/pytest_${TAG}_$i.log | tail -1)"
  for cfg in ${CONFIGS:-c2 c3 hot}; do
    extra=""; [ $cfg = c3 ] && extra="--config c3 --steps 15"; [ $cfg = hot ] && extra="--hot 0.02"
    env $v timeout 300 $B $extra > $OUT/bench_${TAG}_${i}_$cfg.json 2>> $OUT/bench_${TAG}.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${TAG}_${i}_$cfg.json").read().strip().splitlines()[-1])
    print("[$v] $cfg", round(d["value"],1), "/s", round(d["ms_per_step"],4), "ms", {k:round(v,4) for k,v in d["kernel_ms"].items()}, {k:round(v,3) for k,v in d["stage_ms"].items()})
except Exception as e:
    print("[$v] $cfg", "FAILED", e)
PY
  done
  if [ "${STATS:-0}" = 1 ]; then
    for c in ${STATS_CONFIGS:-C2}; do
      env $v GSPLAT_HIP_LIB=$ROOT/opensplat_amd/csrc/libgsplat_hip_stats.so timeout 300 python scripts/work_stats.py $c > $OUT/work_stats_${TAG}_${i}_$c.json 2>> $OUT/bench_${TAG}.err
      echo "[$v] stats $c: $(cat $OUT/work_stats_${TAG}_${i}_$c.json)"
    done
  fi
  i=$((i+1))
done
if [ -n "${AB_FLAGS:-}" ]; then
  for cfg in ${AB_CONFIGS:-c2 c3}; do
    timeout 600 python scripts/ab_raster.py $cfg $AB_FLAGS 2>&1 | grep flags | tee $OUT/ab_raster_${TAG}_$cfg.log
  done
fi
