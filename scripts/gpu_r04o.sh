#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_segmented.py tests/test_gpu_model_fused.py tests/test_gpu_train.py -x -q 2>&1 | tail -30
