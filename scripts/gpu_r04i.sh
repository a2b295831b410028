#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_segmented.py -q -x 2>&1 | tail -30 > $OUT/pytest_seg.log; cat $OUT/pytest_seg.log
for S in 64 256 512; do
echo "seg_len $S"
GSPLAT_SEG_LEN=$S timeout 300 python scripts/timeline_sweep.py 384 288 2>&1 >/dev/null | grep gaussians | grep -o "'gaussians': [0-9]*\|k_rasterize_backward.*"
GSPLAT_SEG_LEN=$S timeout 300 python scripts/timeline_sweep.py 96 72 2>&1 >/dev/null | grep gaussians | grep -o "'gaussians': [0-9]*\|k_rasterize_backward.*"
done
timeout 600 python scripts/train_synthetic.py --no-cpu > $OUT/e2e_synthetic_r04_seg.json 2> $OUT/e2e_seg.err; tail -c 900 $OUT/e2e_synthetic_r04_seg.json
