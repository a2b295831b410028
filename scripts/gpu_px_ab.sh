#!/bin/bash
# same-box A/B of the backward's pixels per lane: heuristic (default) vs forced 2 vs forced 4
cd ${GRAFT_REPO_ROOT:-/root/repo}
B="python bench.py --no-cpu-baseline --steps 40 --warmup 5"
for a in "" "--hot 0.02" "--config c3 --steps 15" "--config c4-sequence --steps 48 --warmup 8"; do
  for px in "" 2 4; do
    GSPLAT_BWD_PX=$px $B $a 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$a] px=$px', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()})"
  done
done
