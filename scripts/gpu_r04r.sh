#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_r04r.log 2>&1; grep -n "passed\|failed" $OUT/pytest_r04r.log | tail -3; grep -n "^E  " $OUT/pytest_r04r.log | head -10
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_r04r.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("$OUT/bench_r04r.json").read().strip().splitlines()[-1]); print("C2", round(d["value"],1), {k:round(v*1e3,1) for k,v in d["kernel_ms"].items()})
PY
timeout 600 python scripts/train_synthetic.py --no-cpu --no-segments > $OUT/e2e_noseg.json 2> $OUT/e2e_noseg.err
timeout 600 python scripts/train_synthetic.py --no-cpu > $OUT/e2e_seg.json 2> $OUT/e2e_seg.err
timeout 600 python scripts/train_synthetic.py --no-cpu --no-segments > $OUT/e2e_noseg2.json 2> $OUT/e2e_noseg.err
timeout 600 python scripts/train_synthetic.py --no-cpu > $OUT/e2e_seg2.json 2> $OUT/e2e_seg.err
python - <<PY
import json
for f in ("e2e_noseg","e2e_seg","e2e_noseg2","e2e_seg2"):
    d=json.load(open("$OUT/%s.json"%f)); print(f, round(d["iterations_per_s"],1), "it/s", d["final_gaussians"], {k:v for k,v in d.items() if "psnr" in k.lower()})
PY
timeout 300 python scripts/bench_model_fused.py > $OUT/model_fused_seg.json 2> $OUT/model_fused_seg.err; tail -c 500 $OUT/model_fused_seg.json
