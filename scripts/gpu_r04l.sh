#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
for wh in "752 500" "1008 756" "1504 1000"; do
for mode in plain seg64 seg128; do
  case $mode in
    plain) export GSPLAT_SEG_LEN=; unset GSPLAT_SEG_LEN; unset GSPLAT_SEG_FORCE;;
    seg64) export GSPLAT_SEG_LEN=64 GSPLAT_SEG_FORCE=1;;
    seg128) export GSPLAT_SEG_LEN=128 GSPLAT_SEG_FORCE=1;;
  esac
  echo "== $wh $mode"
  timeout 300 python scripts/timeline_sweep.py $wh 20000,100000 2>&1 >/dev/null | grep gaussians | grep -o "'gaussians': [0-9]*\|'M': [0-9.]*\|'longest_list': [0-9.]*\|k_rasterize.*" | tr '\n' ' '; echo
done
done
