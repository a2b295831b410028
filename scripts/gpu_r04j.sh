#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
for seg in 0 1; do
echo "segmented $seg"
GSPLAT_SEGMENTED=$seg GSPLAT_SEG_LEN=64 timeout 300 python scripts/timeline_small.py 6000 384 288 2>&1 >/dev/null | grep -v amdgpu
GSPLAT_SEGMENTED=$seg GSPLAT_SEG_LEN=64 timeout 300 python scripts/timeline_small.py 6000 96 72 2>&1 >/dev/null | tail -1
done
