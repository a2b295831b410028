#!/bin/bash
# Several cameras per step: the two-in-flight test, bench lines with 2 / 4 cameras per step on one GPU (serial loop
# beside it), two ranks over gloo on the one GPU (exercises cameras_c1_c2 / communicator of the N > 1 line).
set -u
TAG=${1:-cams}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 900 python -m pytest tests/test_gpu_two_in_flight.py ${EXTRA_TESTS:-} -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1
echo "pytest rc=$? $(grep -E 'passed|failed|error' $OUT/pytest_$TAG.log | tail -1)"
for c in 2 4; do
  timeout 300 python bench.py --no-cpu-baseline --cameras-per-rank $c --steps 20 --warmup 5 > $OUT/bench_${TAG}_cpr$c.json 2>> $OUT/bench_$TAG.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${TAG}_cpr$c.json").read().strip().splitlines()[-1])
    print("cpr$c", round(d["value"],1), "/s", round(d["ms_per_step"],4), "ms", d["cameras_ab"])
except Exception as e:
    print("cpr$c FAILED", e)
PY
done
if [ "${GLOO2:-1}" = 1 ]; then
GSPLAT_DIST_BACKEND=gloo timeout 600 python bench.py --no-cpu-baseline --gpus 2 --steps 6 --warmup 2 > $OUT/bench_${TAG}_gloo2.json 2>> $OUT/bench_$TAG.err
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${TAG}_gloo2.json").read().strip().splitlines()[-1])
    print("gloo2", round(d["value"],1), "/s", d["cameras_c1_c2"], d["communicator"], d["per_link_bytes_per_step"], d["exchange_ab"])
except Exception as e:
    print("gloo2 FAILED", e)
PY
fi
tail -5 $OUT/bench_$TAG.err
