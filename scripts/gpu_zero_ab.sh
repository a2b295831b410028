#!/bin/bash
# A/B: the gradient records zeroed by the binning's count pass (gs_bin_speculative_zero) vs the fill kernel in front of the
# compositing backward (GSPLAT_RECORDS_MEMSET=1).  Same box, interleaved.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
for m in 0 1; do
  GSPLAT_RECORDS_MEMSET=$m python bench.py --no-cpu-baseline --steps 40 --warmup 5 $EXTRA 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ks={k['kernel']:round(k['ms']*1000,1) for k in d['kernels'] if 'kernel' in k and 'stage' not in k and ('count' in k['kernel'] or 'memset' in k['kernel'])}
print('memset=$m', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()}, ks, {k:round(v,3) for k,v in d['stage_ms'].items() if k in ('bin_sort','rasterize_bwd')})"
done
done
