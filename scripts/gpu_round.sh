#!/bin/bash
# One complete evidence run for a round: -m gpu suite, bench lines (C2 default incl. cpu_baseline, C3,
# fast-exp, moving camera, hot spot, 2 / 4 cameras per step on one GPU, 2 ranks over gloo on the one GPU, the
# round-4 backward kernels beside the default), work counters, rocprofv3 passes, training step, end to end.
#   gpu_round.sh TAG        LEAN=1: no fast-exp / Morton / c4-sequence / gloo lines, no COLMAP e2e run, C3 kernel stats
#                           without PMC passes (unless C3_PMC=1)
#                           SMALL_FRAMES=1: also the frames-of-few-tiles measurements of DESIGN.md 4.3
set -u
TAG=${1:-r05}
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
timeout 300 $B --hot 0.02 > $OUT/bench_${TAG}_hot.json 2>> $OUT/bench_$TAG.err
timeout 300 $B --cameras-per-rank 2 > $OUT/bench_${TAG}_cpr2.json 2>> $OUT/bench_$TAG.err
# the four-group backward kernels of rounds 2 - 4 on the same box (flag bits 25..26 = 2)
GSPLAT_BWD_FLAGS=0x4000000 timeout 300 $B > $OUT/bench_${TAG}_classic.json 2>> $OUT/bench_$TAG.err
GSPLAT_BWD_FLAGS=0x4000000 timeout 300 $B --config c3 --steps 20 > $OUT/bench_${TAG}_classic_c3.json 2>> $OUT/bench_$TAG.err
if [ "${LEAN:-0}" != 1 ]; then
timeout 300 $B --fast-exp > $OUT/bench_${TAG}_fastexp.json 2>> $OUT/bench_$TAG.err
timeout 300 $B --order morton > $OUT/bench_${TAG}_morton.json 2>> $OUT/bench_$TAG.err
timeout 300 $B --cameras-per-rank 4 > $OUT/bench_${TAG}_cpr4.json 2>> $OUT/bench_$TAG.err
timeout 300 $B --config c4-sequence --steps 48 --warmup 0 > $OUT/bench_${TAG}_c4seq_cold.json 2>> $OUT/bench_$TAG.err
timeout 300 $B --config c4-sequence --steps 48 --warmup 8 > $OUT/bench_${TAG}_c4seq.json 2>> $OUT/bench_$TAG.err
GSPLAT_DIST_BACKEND=gloo timeout 600 $B --gpus 2 --steps 10 --warmup 2 > $OUT/bench_${TAG}_gloo2.json 2>> $OUT/bench_$TAG.err
GSPLAT_DIST_BACKEND=gloo timeout 600 $B --gpus 2 --steps 10 --warmup 2 --exchange flat > $OUT/bench_${TAG}_gloo2_flat.json 2>> $OUT/bench_$TAG.err
GSPLAT_DIST_BACKEND=gloo timeout 600 $B --gpus 2 --steps 10 --warmup 2 --cameras-per-rank 4 > $OUT/bench_${TAG}_gloo2_c4.json 2>> $OUT/bench_$TAG.err
fi
# work counters of the compositing kernels (instrumented build: scripts/build_variant.sh stats -DGS_STATS)
for c in C2 C3; do
lc=$(echo $c | tr A-Z a-z)
GSPLAT_HIP_LIB=$ROOT/opensplat_amd/csrc/libgsplat_hip_stats.so timeout 300 python scripts/work_stats.py $c > $OUT/work_stats_${TAG}_$lc.json 2>> $OUT/bench_$TAG.err
done
bash scripts/profile.sh $TAG > /dev/null 2>&1
if [ "${LEAN:-0}" = 1 ] && [ "${C3_PMC:-0}" != 1 ]; then
  TRACE_ONLY=1 BENCH_ARGS="--config c3" bash scripts/profile.sh ${TAG}_c3 > /dev/null 2>&1
else
  BENCH_ARGS="--config c3" bash scripts/profile.sh ${TAG}_c3 > /dev/null 2>&1
fi
# the training iteration at C2 size (row f2) and the end-to-end synthetic runs (stand-in for config 5)
timeout 600 python scripts/bench_train_step.py > $OUT/f2_train_step_$TAG.json 2>> $OUT/bench_$TAG.err
timeout 600 python scripts/train_synthetic.py > $OUT/e2e_synthetic_$TAG.json 2>> $OUT/bench_$TAG.err
if [ "${LEAN:-0}" != 1 ]; then
timeout 600 python scripts/train_synthetic.py --via-colmap > $OUT/e2e_colmap_$TAG.json 2>> $OUT/bench_$TAG.err
fi
if [ "${SMALL_FRAMES:-0}" = 1 ]; then
  # frames that do not fill the chip: list-length sweeps one pass / pieces, the per-kernel
  # timeline of a small training iteration, accuracy of the pieces
  for wh in "96 72" "384 288" "1008 756"; do
    t=$(echo $wh | tr ' ' 'x')
    GSPLAT_SEGMENTED=0 timeout 300 python scripts/timeline_sweep.py $wh > $OUT/timeline_sweep_${t}_onepass_$TAG.json 2>/dev/null
    GSPLAT_SEGMENTED=1 timeout 300 python scripts/timeline_sweep.py $wh > $OUT/timeline_sweep_${t}_pieces_$TAG.json 2>/dev/null
  done
  GSPLAT_SEGMENTED=1 timeout 300 python scripts/timeline_small.py 6000 384 288 > $OUT/timeline_small_6000_$TAG.json 2>/dev/null
  timeout 300 python scripts/diag_seg_accuracy.py > $OUT/seg_accuracy_$TAG.json 2>/dev/null
fi
for f in "" _c3 _hot _cpr2 _classic _classic_c3 _fastexp _morton _cpr4 _c4seq_cold _c4seq _gloo2 _gloo2_flat _gloo2_c4; do python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_${TAG}$f.json").read().strip().splitlines()[-1])
    print("${TAG}$f", round(d["value"],1), "/s", round(d["ms_per_step"],4), "ms", {k:round(v,4) for k,v in d["kernel_ms"].items()}, {k:round(v,3) for k,v in d["stage_ms"].items()}, d.get("speculative_binning"))
except Exception as e:
    print("${TAG}$f", "FAILED", e)
PY
done
cat $OUT/work_stats_${TAG}_c2.json
