#!/bin/bash
# How the tables of DESIGN.md §4.3 / profiles/HISTORY.md ("frames that do not fill the chip") were measured: the
# compositing kernels' durations on SfM-like training scenes (scripts/timeline_sweep.py) by frame size, with the
# one-pass backward, with the library's own plan, and with forced piece lengths / pixels per lane; the forward with
# one / two entries per step; bench.py's C2 / C3 / hot-spot lines with forced pieces.  Measurement knobs (read by
# opensplat_amd/cabi.py and bench.py, not by the library):
#   GSPLAT_SEGMENTED=0        Trainer(segmented=False)
#   GSPLAT_SEG_LEN=S          piece length instead of the plan's  (+ GSPLAT_SEG_FORCE=1: also where the plan says no)
#   GSPLAT_BWD_FLAGS=0x200000 / 0x400000 / 0x600000   1 / 2 / 4 pixels per lane (flag bits 21..22)
#   GSPLAT_FWD_FLAGS=0x800000 / 0x1000000             one / two entries per forward step (flag bits 23..24)
#   GSPLAT_BENCH_PIECES=S     bench.py: checkpointed forward + four-pixel pieces of S entries
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
run() { timeout 300 python scripts/timeline_sweep.py $1 $2 $3 2>&1 >/dev/null | grep gaussians | grep -o "'gaussians': [0-9]*\|'longest_list': [0-9.]*\|k_rasterize[^:]*: [0-9.]*" | tr '\n' ' '; echo; }
for wh in "96 72" "384 288" "640 480" "752 500" "1008 756" "1504 1000" "1920 1080"; do
  unset GSPLAT_SEG_LEN GSPLAT_SEG_FORCE GSPLAT_BWD_FLAGS GSPLAT_FWD_FLAGS
  echo "== $wh one pass"; GSPLAT_SEGMENTED=0 run $wh 6000,100000
  echo "== $wh library plan"; run $wh 6000,100000
  for px in 0x200000 0x400000 0x600000; do for S in 64 128; do
    echo "== $wh pieces of $S, pixel flag $px"; GSPLAT_SEG_LEN=$S GSPLAT_SEG_FORCE=1 GSPLAT_BWD_FLAGS=$px run $wh 6000,100000
  done; done
  for fl in 0x800000 0x1000000; do echo "== $wh forward flags $fl"; GSPLAT_SEGMENTED=0 GSPLAT_FWD_FLAGS=$fl run $wh 6000,100000; done
done
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', round(d['value'],1), '/s', {k:round(v*1e3,1) for k,v in d['kernel_ms'].items()})"; }
for P in 0 64 128 256; do GSPLAT_BENCH_PIECES=$P timeout 300 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | show "C2 pieces=$P"; done
for P in 0 128 256; do GSPLAT_BENCH_PIECES=$P timeout 300 python bench.py --no-cpu-baseline --config c3 --steps 30 2>/dev/null | show "C3 pieces=$P"; done
for P in 0 64 128; do GSPLAT_BENCH_PIECES=$P timeout 300 python bench.py --no-cpu-baseline --hot 0.02 --steps 200 2>/dev/null | show "C2 hot spot pieces=$P"; done
