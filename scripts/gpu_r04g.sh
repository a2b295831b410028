#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python scripts/bench_train_graph.py > $OUT/train_graph_r04.json 2> $OUT/train_graph_r04.err; tail -6 $OUT/train_graph_r04.err
for rep in 1 2 3; do
timeout 600 python scripts/train_synthetic.py --graph > $OUT/e2e_synthetic_r04g_graph$rep.json 2> $OUT/e2e_r04g_graph$rep.err; tail -2 $OUT/e2e_r04g_graph$rep.err
done
timeout 600 python scripts/train_synthetic.py > $OUT/e2e_synthetic_r04g_plain.json 2> $OUT/e2e_r04g_plain.err
python - <<PY
import json
for n in ("graph1", "graph2", "graph3", "plain"):
    try:
        d = json.loads(open("$OUT/e2e_synthetic_r04g_%s.json" % n).read().strip().splitlines()[-1])
        print(n, round(d["iterations_per_s"], 1), "it/s", "psnr", [round(c["psnr"], 2) for c in d["psnr_curve"]], "N", d["final_gaussians"], d.get("captured_iterations"))
    except Exception as e:
        print(n, "FAILED", e)
PY
