#!/bin/bash
# quick GPU check: compositing / parity tests + C2 / C3 bench lines
set -u
TAG=${1:-q}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deterministic.py tests/test_gpu_baseline_parity.py tests/test_gpu_ops_and_edges.py -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_$TAG.log
tail -4 $OUT/pytest_$TAG.log
B="python bench.py --no-cpu-baseline --steps 40 --warmup 5"
timeout 300 $B ${BENCH_EXTRA:-} > $OUT/bench_${TAG}_c2.json 2> $OUT/bench_${TAG}.err
timeout 300 $B --config c3 --steps 15 ${BENCH_EXTRA:-} > $OUT/bench_${TAG}_c3.json 2>> $OUT/bench_${TAG}.err
for f in c2 c3; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${TAG}_$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"],1), "it/s", round(d["ms_per_step"],4), "ms", {k:round(v,4) for k,v in d["kernel_ms"].items()}, {k:round(v,3) for k,v in d["stage_ms"].items()})
except Exception as e:
    print("$f", "FAILED", e)
PY
done
