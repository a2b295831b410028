#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_r04n.log 2>&1; grep -n "passed\|failed" $OUT/pytest_r04n.log | tail -3; grep -n "Error\|assert" $OUT/pytest_r04n.log | head -20
timeout 300 python scripts/bench_model_fused.py > $OUT/model_fused_seg.json 2> $OUT/model_fused_seg.err; tail -c 600 $OUT/model_fused_seg.json
