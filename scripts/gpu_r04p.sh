#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 600 python scripts/diag_seg_accuracy.py > $OUT/seg_accuracy_r04.json 2> $OUT/seg_accuracy.err; grep -v amdgpu $OUT/seg_accuracy.err | tail -20
