#!/bin/bash
# A/B of differently built libgsplat_hip variants on one box: parity tests on the default library,
# then the C2 / C3 bench lines per variant (GSPLAT_HIP_LIB selects the library).
#   usage: gpu_variants.sh TAG suffix1 suffix2 ...    ("" = the default library)
set -u
TAG=${1:-v}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deterministic.py tests/test_gpu_baseline_parity.py tests/test_gpu_ops_and_edges.py -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_$TAG.log
tail -4 $OUT/pytest_$TAG.log
B="python bench.py --no-cpu-baseline --steps 40 --warmup 5"
for v in default "$@"; do
  if [ "$v" = default ]; then unset GSPLAT_HIP_LIB; else export GSPLAT_HIP_LIB=$ROOT/opensplat_amd/csrc/libgsplat_hip_$v.so; fi
  for rep in 1 2; do
    timeout 300 $B ${BENCH_EXTRA:-} > $OUT/bench_${TAG}_${v}_c2_$rep.json 2>> $OUT/bench_${TAG}.err
  done
  timeout 300 $B --config c3 --steps 15 ${BENCH_EXTRA:-} > $OUT/bench_${TAG}_${v}_c3_1.json 2>> $OUT/bench_${TAG}.err
  for f in c2_1 c2_2 c3_1; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${TAG}_${v}_$f.json").read().strip().splitlines()[-1])
    print("$v $f", round(d["value"],1), "it/s", round(d["ms_per_step"],4), "ms", {k:round(v,4) for k,v in d["kernel_ms"].items()}, {k:round(v,3) for k,v in d["stage_ms"].items()})
except Exception as e:
    print("$v $f", "FAILED", e)
PY
  done
done
