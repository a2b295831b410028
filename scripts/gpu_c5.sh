#!/bin/bash
# A config-5-SHAPED run (BASELINE config 5 itself — the banana capture — is not available offline): a COLMAP
# project on disk, 24 + 4 cameras at 1008x756, 50 000 sparse points, `opensplat -n 7000` schedules as they are,
# PSNR curve + wall clock, with the reference's CPU chain timed on the same host for one iteration per resolution.
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
( time timeout 900 python scripts/train_synthetic.py --via-colmap --iters 7000 --width 1008 --height 756 --gt-gaussians 200000 --init-points 50000 --reference-schedules --cpu-baseline ) > gpurun_out/c5_standin_${TAG:-r05}.json 2> gpurun_out/c5_standin_${TAG:-r05}.err
tail -4 gpurun_out/c5_standin_${TAG:-r05}.err
python - <<PY
import json
d=json.loads(open("gpurun_out/c5_standin_${TAG:-r05}.json").read().strip().splitlines()[-1])
print(round(d["iterations_per_s"],1), "it/s", d["train_seconds"], "s", [round(c["psnr"],2) for c in d["psnr_curve"]], d["final_gaussians"], d.get("cpu_baseline"))
PY
