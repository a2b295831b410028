#!/bin/bash
# what the driver runs at round end, on the committed tree: smoke, the -m gpu suite, the default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_final.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; tail -4 gpurun_out/bench_final.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["sustained"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["roofline_valu"][0]["source"][:60])
PY
timeout 600 python scripts/bench_model_fused.py > gpurun_out/model_fused_r04.json 2> gpurun_out/model_fused_r04.err; tail -2 gpurun_out/model_fused_r04.err; head -c 600 gpurun_out/model_fused_r04.json
