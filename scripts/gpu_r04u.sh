#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_segmented.py -x -q 2>&1 | tail -3
run() { timeout 300 python scripts/timeline_sweep.py $1 $2 $3 2>&1 >/dev/null | grep gaussians | grep -o "'gaussians': [0-9]*\|k_rasterize[^:]*: [0-9.]*" | tr '\n' ' '; echo; }
unset GSPLAT_SEG_LEN GSPLAT_SEG_FORCE GSPLAT_BWD_FLAGS
echo "== 384 288 default"; run 384 288 6000
for wh in "640 480" "752 500" "1008 756"; do
  echo "== $wh plain"; unset GSPLAT_SEG_LEN GSPLAT_SEG_FORCE GSPLAT_BWD_FLAGS; GSPLAT_SEGMENTED=0; export GSPLAT_SEG_LEN=100000 GSPLAT_SEG_FORCE=1; run $wh 20000,100000
  for px in 0x200000 0x400000 0x600000; do for S in 64 128; do
  echo "== $wh pieces px-flag $px S=$S"; export GSPLAT_SEG_LEN=$S GSPLAT_SEG_FORCE=1 GSPLAT_BWD_FLAGS=$px; run $wh 20000,100000
  done; done
done
echo "== 1504 1000 PX4 S=128"; export GSPLAT_SEG_LEN=128 GSPLAT_SEG_FORCE=1 GSPLAT_BWD_FLAGS=0x600000; run 1504 1000 100000,1000000
