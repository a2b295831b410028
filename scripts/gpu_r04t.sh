#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 600 python scripts/diag_seg_accuracy.py 2>&1 >/dev/null | grep -v amdgpu | cut -c1-330 | tail -3
timeout 600 python -m pytest tests/test_gpu_segmented.py -x -q 2>&1 | tail -3
run() { timeout 300 python scripts/timeline_sweep.py $1 $2 $3 2>&1 >/dev/null | grep gaussians | grep -o "'gaussians': [0-9]*\|k_rasterize[^:]*: [0-9.]*" | tr '\n' ' '; echo; }
echo "== 384 288"; run 384 288 6000
for wh in "1008 756" "1504 1000" "1920 1080"; do
  echo "== $wh plain"; unset GSPLAT_SEG_LEN GSPLAT_SEG_FORCE GSPLAT_BWD_FLAGS; run $wh 100000,1000000
  for S in 64 128; do
  echo "== $wh pieces PX4 S=$S"; export GSPLAT_SEG_LEN=$S GSPLAT_SEG_FORCE=1 GSPLAT_BWD_FLAGS=0x600000; run $wh 100000,1000000
  done
done
