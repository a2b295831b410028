#!/bin/bash
# round 4, call B: the tests that failed / are new after call A, direct-vs-moments position gradients A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_reference_signature.py tests/test_gpu_launcher_level.py tests/test_integration_build.py tests/test_gpu_c3_golden.py tests/test_gpu_dist_rccl.py tests/test_gpu_parity.py tests/test_gpu_deterministic.py -m gpu -q --durations=5 > $OUT/pytest_r04b.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_r04b.log
tail -40 $OUT/pytest_r04b.log
timeout 300 $ROOT/oracle/_ref/model_forward_shim --gpu-ordered > $OUT/shim_ordered_r04b.log 2>&1; cat $OUT/shim_ordered_r04b.log
# needles with the round-3 moments (the variant library): the failure the direct form fixes
GSPLAT_HIP_LIB=$ROOT/opensplat_amd/csrc/libgsplat_hip_moments.so timeout 300 python -m pytest "tests/test_gpu_reference_signature.py::test_ten_argument_call_matches_the_oracle[needles]" -m gpu -q 2>&1 | grep -E "AssertionError: assert|passed|failed" | tee $OUT/needles_moments_r04b.log
REPS="1 2" bash scripts/gpu_lib_ab.sh moments 2>&1 | tee $OUT/ab_directxy_r04b.log
