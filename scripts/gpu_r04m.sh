#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
for wh in "640 480" "752 500" "1008 756" "1504 1000"; do
for fl in 0x800000 0x1000000; do
  echo "== $wh fwd flags $fl"
  GSPLAT_FWD_FLAGS=$fl timeout 300 python scripts/timeline_sweep.py $wh 20000,100000 2>&1 >/dev/null | grep gaussians | grep -o "'gaussians': [0-9]*\|k_rasterize_forward<[^>]*>': [0-9.]*" | tr '\n' ' '; echo
done
done
