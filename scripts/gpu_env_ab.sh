#!/bin/bash
# A/B of environment-selected variants on one box:  gpu_env_ab.sh TAG "VAR=val" "VAR2=val" ...
set -u
TAG=${1:-e}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
B="python bench.py --no-cpu-baseline --steps 40 --warmup 5"
i=0
for v in "" "$@"; do
  for cfg in c2 c3; do
    extra=""; [ $cfg = c3 ] && extra="--config c3 --steps 15"
    env $v timeout 300 $B $extra > $OUT/bench_${TAG}_${i}_$cfg.json 2>> $OUT/bench_${TAG}.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${TAG}_${i}_$cfg.json").read().strip().splitlines()[-1])
    print("[$v] $cfg", round(d["value"],1), "it/s", round(d["ms_per_step"],4), "ms", {k:round(v,4) for k,v in d["kernel_ms"].items()}, {k:round(v,3) for k,v in d["stage_ms"].items()})
except Exception as e:
    print("[$v] $cfg", "FAILED", e)
PY
  done
  i=$((i+1))
done
