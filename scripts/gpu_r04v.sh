#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), '/s', {k:round(v*1e3,1) for k,v in d['kernel_ms'].items()})"; }
for P in 0 64 128 256; do
GSPLAT_BENCH_PIECES=$P timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | show "C2 pieces=$P"
done
for P in 0 128 256; do
GSPLAT_BENCH_PIECES=$P timeout 300 python bench.py --no-cpu-baseline --config c3 --steps 30 2>/dev/null | show "C3 pieces=$P"
done
for P in 0 64 128; do
GSPLAT_BENCH_PIECES=$P timeout 300 python bench.py --no-cpu-baseline --hot 0.02 --steps 200 2>/dev/null | show "C2hot pieces=$P"
done
run() { timeout 300 python scripts/timeline_sweep.py $1 $2 $3 2>&1 >/dev/null | grep gaussians | grep -o "'gaussians': [0-9]*\|k_rasterize[^:]*: [0-9.]*" | tr '\n' ' '; echo; }
for px in 0x200000 0x400000 0x600000; do
echo "== 384 288 / 96 72 px-flag $px"; export GSPLAT_SEG_LEN=64 GSPLAT_SEG_FORCE=1 GSPLAT_BWD_FLAGS=$px; run 384 288 6000,20000; run 96 72 6000
done
