#!/bin/bash
# One complete evidence run for a round: -m gpu suite, bench lines (C2 default incl. cpu_baseline, C3,
# fast-exp, moving camera, hot spot, 2 ranks over gloo on the one GPU), work counters, rocprofv3 passes.
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke_$TAG.log 2>&1; tail -1 $OUT/smoke_$TAG.log
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/pytest_$TAG.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_$TAG.log
tail -14 $OUT/pytest_$TAG.log
timeout 300 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
B="python bench.py --no-cpu-baseline"
timeout 300 $B --config c3 --steps 20 > $OUT/bench_${TAG}_c3.json 2>> $OUT/bench_$TAG.err
timeout 300 $B --fast-exp > $OUT/bench_${TAG}_fastexp.json 2>> $OUT/bench_$TAG.err
timeout 300 $B --hot 0.02 > $OUT/bench_${TAG}_hot.json 2>> $OUT/bench_$TAG.err
timeout 300 $B --order morton > $OUT/bench_${TAG}_morton.json 2>> $OUT/bench_$TAG.err
timeout 300 $B --config c4-sequence --steps 48 --warmup 0 > $OUT/bench_${TAG}_c4seq_cold.json 2>> $OUT/bench_$TAG.err
timeout 300 $B --config c4-sequence --steps 48 --warmup 8 > $OUT/bench_${TAG}_c4seq.json 2>> $OUT/bench_$TAG.err
GSPLAT_DIST_BACKEND=gloo timeout 600 $B --gpus 2 --steps 10 --warmup 2 > $OUT/bench_${TAG}_gloo2.json 2>> $OUT/bench_$TAG.err
GSPLAT_DIST_BACKEND=gloo timeout 600 $B --gpus 2 --steps 10 --warmup 2 --exchange flat > $OUT/bench_${TAG}_gloo2_flat.json 2>> $OUT/bench_$TAG.err
GSPLAT_DIST_BACKEND=gloo timeout 600 $B --gpus 2 --steps 10 --warmup 2 --cameras-per-rank 4 > $OUT/bench_${TAG}_gloo2_c4.json 2>> $OUT/bench_$TAG.err
# work counters of the compositing kernels (instrumented build: scripts/build_variant.sh stats -DGS_STATS)
for c in C2 C3; do
lc=$(echo $c | tr A-Z a-z)
GSPLAT_HIP_LIB=$ROOT/opensplat_amd/csrc/libgsplat_hip_stats.so timeout 300 python scripts/work_stats.py $c > $OUT/work_stats_${TAG}_$lc.json 2>> $OUT/bench_$TAG.err
done
bash scripts/profile.sh $TAG > /dev/null 2>&1
BENCH_ARGS="--config c3" bash scripts/profile.sh ${TAG}_c3 > /dev/null 2>&1
# the training iteration at C2 size (row f2) and the end-to-end synthetic runs (stand-in for config 5)
timeout 600 python scripts/bench_train_step.py > $OUT/f2_train_step_$TAG.json 2>> $OUT/bench_$TAG.err
timeout 600 python scripts/train_synthetic.py > $OUT/e2e_synthetic_$TAG.json 2>> $OUT/bench_$TAG.err
timeout 600 python scripts/train_synthetic.py --via-colmap > $OUT/e2e_colmap_$TAG.json 2>> $OUT/bench_$TAG.err
for f in "" _c3 _fastexp _hot _morton _c4seq_cold _c4seq _gloo2 _gloo2_flat _gloo2_c4; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${TAG}$f.json").read().strip().splitlines()[-1])
    print("${TAG}$f", round(d["value"],1), "/s", round(d["ms_per_step"],4), "ms", {k:round(v,4) for k,v in d["kernel_ms"].items()}, {k:round(v,3) for k,v in d["stage_ms"].items()}, d.get("speculative_binning"))
except Exception as e:
    print("${TAG}$f", "FAILED", e)
PY
done
cat $OUT/work_stats_${TAG}_c2.json
