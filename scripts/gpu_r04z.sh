#!/bin/bash
# Round-4 evidence on the final tree: scripts/gpu_round.sh r04 (suite, bench lines, rocprofv3 passes, f2, end to end)
# + what the few-tile work added: list-length sweeps one-pass / pieces, rocprofv3 kernel stats of a small-frame
# training run, accuracy of the pieces, the config-5-shaped run, the C++ fused model loop.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
bash scripts/gpu_round.sh r04 > $OUT/gpu_round_r04.out 2>&1; tail -16 $OUT/gpu_round_r04.out
for wh in "96 72" "384 288" "1008 756"; do
  t=$(echo $wh | tr ' ' 'x')
  GSPLAT_SEGMENTED=0 timeout 300 python scripts/timeline_sweep.py $wh > $OUT/timeline_sweep_${t}_onepass_r04.json 2>/dev/null
  GSPLAT_SEGMENTED=1 timeout 300 python scripts/timeline_sweep.py $wh > $OUT/timeline_sweep_${t}_pieces_r04.json 2>/dev/null
done
GSPLAT_SEGMENTED=0 timeout 300 python scripts/timeline_small.py 6000 384 288 > $OUT/timeline_small_6000_onepass_r04.json 2>/dev/null
GSPLAT_SEGMENTED=1 timeout 300 python scripts/timeline_small.py 6000 384 288 > $OUT/timeline_small_6000_r04.json 2>/dev/null
GSPLAT_SEGMENTED=1 timeout 300 python scripts/timeline_small.py 6000 96 72 > $OUT/timeline_small_6000_lowres_r04.json 2>/dev/null
timeout 300 python scripts/diag_seg_accuracy.py > $OUT/seg_accuracy_r04.json 2>/dev/null
timeout 600 python scripts/train_synthetic.py --no-cpu --no-segments > $OUT/e2e_synthetic_r04_onepass.json 2>/dev/null
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_small -o small -- python $ROOT/scripts/train_synthetic.py --no-cpu --iters 1500 > $OUT/prof_small.log 2>&1 )
find $OUT/prof_small -name "*kernel_stats.csv" | head -2
bash scripts/gpu_c5.sh 2>&1 | tail -2
timeout 600 python scripts/bench_model_fused.py > $OUT/model_fused_r04.json 2>/dev/null; head -c 500 $OUT/model_fused_r04.json; echo
python - <<PY
import json
for f in ("e2e_synthetic_r04_onepass","e2e_synthetic_r04","e2e_colmap_r04"):
    try:
        d=json.load(open("$OUT/%s.json"%f)); print(f, round(d["iterations_per_s"],1), "it/s", d["train_seconds"], d["final_gaussians"], round(d["psnr_curve"][-1]["psnr"],2))
    except Exception as e: print(f, "FAILED", e)
PY
