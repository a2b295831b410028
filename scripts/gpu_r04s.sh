#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
run() { timeout 300 python scripts/timeline_sweep.py $1 $2 $3 2>&1 >/dev/null | grep gaussians | grep -o "'gaussians': [0-9]*\|'longest_list': [0-9.]*\|k_rasterize[^:]*: [0-9.]*" | tr '\n' ' '; echo; }
for wh in "1008 756" "1504 1000"; do
  echo "== $wh plain"; unset GSPLAT_SEG_LEN GSPLAT_SEG_FORCE GSPLAT_BWD_FLAGS; run $wh 20000,100000,400000
  for S in 128 256 512; do
    echo "== $wh pieces PX4 S=$S"; export GSPLAT_SEG_LEN=$S GSPLAT_SEG_FORCE=1 GSPLAT_BWD_FLAGS=0x600000; run $wh 20000,100000,400000
  done
  echo "== $wh pieces PX2 S=128"; export GSPLAT_SEG_LEN=128 GSPLAT_SEG_FORCE=1 GSPLAT_BWD_FLAGS=0x400000; run $wh 20000,100000,400000
done
