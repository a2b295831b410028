#!/bin/bash
# round 6, session 3: queue words as record offsets (forward), one record stride (backward q) against HEAD (base)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06s3
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deterministic.py tests/test_gpu_visibility_threshold.py tests/test_gpu_baseline_parity.py tests/test_gpu_parity_r03.py tests/test_gpu_segmented.py tests/test_gpu_forward_compact.py tests/test_gpu_c3_golden.py -x -q -m gpu 2>&1 | tail -5
REPS="1 2 3" bash scripts/gpu_lib_ab.sh base 2>&1 | tee gpurun_out/r06s3/ab_stride.log
BENCH_EXTRA="--config c3 --steps 20" REPS="1 2" bash scripts/gpu_lib_ab.sh base 2>&1 | tee gpurun_out/r06s3/ab_stride_c3.log
