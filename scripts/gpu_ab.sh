#!/bin/bash
# A/B GPU call: -m gpu suite, then bench lines (backward with 1 / 2 / 4 pixels per lane) at C2 / C3.
set -u
TAG=${1:-ab}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_$TAG.log
tail -4 $OUT/pytest_$TAG.log
B="python bench.py --no-cpu-baseline --steps 40 --warmup 5"
timeout 300 $B > $OUT/bench_${TAG}_c2.json 2> $OUT/bench_${TAG}.err
timeout 300 $B --fast-exp > $OUT/bench_${TAG}_c2_fastexp.json 2>> $OUT/bench_${TAG}.err
timeout 300 $B --config c3 --steps 15 > $OUT/bench_${TAG}_c3.json 2>> $OUT/bench_${TAG}.err
for px in 1 2 4; do
GSPLAT_BWD_PX=$px timeout 300 $B > $OUT/bench_${TAG}_c2_px$px.json 2>> $OUT/bench_${TAG}.err
GSPLAT_BWD_PX=$px timeout 300 $B --config c3 --steps 15 > $OUT/bench_${TAG}_c3_px$px.json 2>> $OUT/bench_${TAG}.err
GSPLAT_BWD_PX=$px timeout 300 $B --hot 0.02 > $OUT/bench_${TAG}_hot_px$px.json 2>> $OUT/bench_${TAG}.err
done
timeout 300 $B --hot 0.02 > $OUT/bench_${TAG}_hot.json 2>> $OUT/bench_${TAG}.err
for f in c2 c2_fastexp c3 hot c2_px1 c2_px2 c2_px4 c3_px1 c3_px2 c3_px4 hot_px1 hot_px2 hot_px4; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${TAG}_$f.json").read().strip().splitlines()[-1])
    print("$f", round(d["value"],1), "it/s", round(d["ms_per_step"],4), "ms", {k:round(v,4) for k,v in d["kernel_ms"].items()}, {k:round(v,3) for k,v in d["stage_ms"].items()})
except Exception as e:
    print("$f", "FAILED", e)
PY
done
