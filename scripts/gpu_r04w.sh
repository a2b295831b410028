#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_r04w.log 2>&1; grep -n "passed\|failed" $OUT/pytest_r04w.log | tail -3; grep -n "^E  " $OUT/pytest_r04w.log | head -10
run() { timeout 300 python scripts/timeline_sweep.py $1 $2 $3 2>&1 >/dev/null | grep gaussians | grep -o "'gaussians': [0-9]*\|k_rasterize[^:]*: [0-9.]*" | tr '\n' ' '; echo; }
for wh in "96 72" "384 288" "752 500" "1008 756" "1504 1000"; do
for seg in 0 1; do echo "== $wh segmented=$seg"; GSPLAT_SEGMENTED=$seg run $wh 6000,100000; done
done
timeout 600 python scripts/train_synthetic.py --no-cpu --no-segments > $OUT/e2e_noseg.json 2> $OUT/e2e_noseg.err
timeout 600 python scripts/train_synthetic.py --no-cpu > $OUT/e2e_seg.json 2> $OUT/e2e_seg.err
python - <<PY
import json
for f in ("e2e_noseg","e2e_seg"):
    d=json.load(open("$OUT/%s.json"%f)); print(f, round(d["iterations_per_s"],1), "it/s", d["final_gaussians"], d["psnr_curve"][-1]["psnr"])
PY
