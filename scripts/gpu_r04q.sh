#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 600 python scripts/diag_seg_accuracy.py > $OUT/seg_accuracy_r04.json 2> $OUT/seg_accuracy.err; grep -v amdgpu $OUT/seg_accuracy.err | cut -c1-600 | tail -6
timeout 600 python -m pytest tests/test_gpu_segmented.py -x -q 2>&1 | tail -15
timeout 300 python scripts/timeline_sweep.py 384 288 6000 2>&1 >/dev/null | grep gaussians
timeout 300 python scripts/timeline_sweep.py 96 72 6000 2>&1 >/dev/null | grep gaussians
