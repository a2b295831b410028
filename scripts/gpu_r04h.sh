#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
timeout 600 python -m pytest tests/test_gpu_segmented.py -q -x 2>&1 | tail -40 > $OUT/pytest_seg.log; cat $OUT/pytest_seg.log
timeout 300 python scripts/timeline_sweep.py 384 288 > $OUT/timeline_sweep_384_seg_r04.json 2> $OUT/ts_a.err
timeout 300 python scripts/timeline_sweep.py 96 72 > $OUT/timeline_sweep_96_seg_r04.json 2> $OUT/ts_b.err
grep -h gaussians $OUT/ts_a.err $OUT/ts_b.err
