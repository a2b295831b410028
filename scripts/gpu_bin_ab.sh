#!/bin/bash
# A/B of the two binnings on one box: GSPLAT_BIN=tiles (rounds 1-5: count / scan / scatter / sort) against the
# strip binning of round 6, at C2, C3 and on the hot-spot scene; the parity test of the two first.
#   gpu_bin_ab.sh TAG
set -u
TAG=${1:-r06a}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_bin_strips.py -x -q > $OUT/pytest_strips_$TAG.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_strips_$TAG.log
tail -25 $OUT/pytest_strips_$TAG.log
GSPLAT_STRIPS_FUSED=1 timeout 900 python -m pytest tests/test_gpu_bin_strips.py -x -q 2>&1 | tail -3
B="python bench.py --no-cpu-baseline"
for mode in ${MODES:-tiles strips tiles strips}; do
  for cfg in c2 c3 hot; do
    args=""; [ $cfg = c3 ] && args="--config c3 --steps 20"; [ $cfg = hot ] && args="--hot 0.02"
    f=$OUT/bench_${TAG}_${mode}_$cfg.json
    [ -s $f ] && f=$OUT/bench_${TAG}_${mode}_${cfg}_2.json
    m=$mode; fu=0; [ $mode = fused ] && m=strips && fu=1
    GSPLAT_STRIPS_FUSED=$fu GSPLAT_BIN=$m timeout 300 $B $args > $f 2>> $OUT/bench_$TAG.err
    python - $f $mode $cfg <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e); sys.exit(0)
ks = {k["kernel"]: round(k["ms"] * 1e3, 1) for k in d.get("kernels", [])}
print(sys.argv[2], sys.argv[3], "value %.1f ms %.4f" % (d["value"], d["ms_per_step"]), "stage", {k: round(v, 4) for k, v in d.get("stage_ms", {}).items()})
print("   ", {k: v for k, v in ks.items() if any(x in k for x in ("count", "scan", "scatter", "sort", "memset", "cell", "strip", "order", "rasterize_f", "rasterize_b")) and "+" not in k})
PY
  done
done
tail -5 $OUT/bench_$TAG.err
