#!/bin/bash
# round 4, call C: log2(e) folded into the staged conic of the backward (one instruction per exponential),
# parity tests on it, A/B against the library without it
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_r03.py tests/test_gpu_reference_signature.py tests/test_gpu_visibility_threshold.py tests/test_gpu_deterministic.py tests/test_gpu_baseline_parity.py tests/test_gpu_ops_and_edges.py tests/test_gpu_c3_golden.py -m gpu -q -x --durations=5 > $OUT/pytest_r04c.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_r04c.log
tail -30 $OUT/pytest_r04c.log
REPS="1 2 3" bash scripts/gpu_lib_ab.sh nol2e 2>&1 | tee $OUT/ab_log2e_r04c.log
