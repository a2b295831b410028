#!/bin/bash
# round 4, call D: the training iteration as one captured HIP graph — tests, then the synthetic end-to-end run both ways
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_train_graph.py tests/test_gpu_train.py -m gpu -q -x --durations=5 > $OUT/pytest_r04d.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_r04d.log
tail -40 $OUT/pytest_r04d.log
timeout 600 python scripts/train_synthetic.py > $OUT/e2e_synthetic_r04d_plain.json 2> $OUT/e2e_r04d.err
timeout 600 python scripts/train_synthetic.py --graph > $OUT/e2e_synthetic_r04d_graph.json 2>> $OUT/e2e_r04d.err
tail -5 $OUT/e2e_r04d.err
python - <<PY
import json
for n in ("plain", "graph"):
    try:
        d = json.loads(open("$OUT/e2e_synthetic_r04d_%s.json" % n).read().strip().splitlines()[-1])
        print(n, round(d["iterations_per_s"], 1), "it/s", "psnr", [round(c["psnr"], 2) for c in d["psnr_curve"]], "N", d["final_gaussians"], d.get("captured_iterations"))
    except Exception as e:
        print(n, "FAILED", e)
PY
