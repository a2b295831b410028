#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
for r in 1 2; do
timeout 600 python scripts/train_synthetic.py --no-cpu --no-segments > $OUT/e2e_noseg.json 2> $OUT/e2e_noseg.err
timeout 600 python scripts/train_synthetic.py --no-cpu > $OUT/e2e_seg.json 2> $OUT/e2e_seg.err
python - <<PY
import json
for f in ("e2e_noseg","e2e_seg"):
    d=json.load(open("$OUT/%s.json"%f)); print(f, round(d["iterations_per_s"],1), "it/s", d.get("psnr_test", d.get("test_psnr")), d["final_gaussians"])
PY
done
