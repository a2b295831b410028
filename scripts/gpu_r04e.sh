#!/bin/bash
# round 4, call E: captured training iteration (tests, end-to-end both ways in one call) + sigma-domain
# thresholds of the backward (parity tests through the variant library, A/B)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_train_graph.py tests/test_gpu_train.py -m gpu -q --durations=5 > $OUT/pytest_r04e.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_r04e.log
tail -40 $OUT/pytest_r04e.log
for rep in 1 2; do
timeout 600 python scripts/train_synthetic.py > $OUT/e2e_synthetic_r04e_plain$rep.json 2> $OUT/e2e_r04e.err
timeout 600 python scripts/train_synthetic.py --graph > $OUT/e2e_synthetic_r04e_graph$rep.json 2>> $OUT/e2e_r04e.err
done
tail -5 $OUT/e2e_r04e.err
python - <<PY
import json
for n in ("plain1", "graph1", "plain2", "graph2"):
    try:
        d = json.loads(open("$OUT/e2e_synthetic_r04e_%s.json" % n).read().strip().splitlines()[-1])
        print(n, round(d["iterations_per_s"], 1), "it/s", "psnr", [round(c["psnr"], 2) for c in d["psnr_curve"]], "N", d["final_gaussians"], d.get("captured_iterations"))
    except Exception as e:
        print(n, "FAILED", e)
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r03.py tests/test_gpu_visibility_threshold.py tests/test_gpu_deterministic.py tests/test_gpu_baseline_parity.py -m gpu -q -x > $OUT/pytest_r04e_main.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_r04e_main.log
tail -4 $OUT/pytest_r04e_main.log
