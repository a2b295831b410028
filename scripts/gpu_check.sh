cd $GRAFT_REPO_ROOT
timeout 1400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
B="python bench.py --no-cpu-baseline --steps 40 --warmup 5"
for a in "" "--hot 0.02" "--config c3 --steps 15" "--config c4-sequence --steps 48 --warmup 8" "--fast-exp"; do
  $B $a 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$a]', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()})"
done
