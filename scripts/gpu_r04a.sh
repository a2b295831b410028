#!/bin/bash
# round 4, call A: suite with the new tests, new bench line, MFMA-vs-DPP backward A/B, ordered-scene shim
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke_r04a.log 2>&1; tail -1 $OUT/smoke_r04a.log
timeout 1500 python -m pytest tests -m gpu -q --durations=8 -x > $OUT/pytest_r04a.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_r04a.log
tail -25 $OUT/pytest_r04a.log
timeout 300 python -m pytest tests/test_gpu_reference_signature.py -m gpu -q > $OUT/pytest_r04a_refsig.log 2>&1; tail -30 $OUT/pytest_r04a_refsig.log
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_r04a.json 2> $OUT/bench_r04a.err; tail -3 $OUT/bench_r04a.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_r04a.json").read().strip().splitlines()[-1])
print("bench", round(d["value"],1), "/s", round(d["ms_per_step"],4), "ms; sustained", d["sustained"], "warmup_run", d["warmup_steps_run"])
for k in d["kernels"]:
    print("  ", k.get("kernel"), k.get("stage",""), round(k["ms"],4), k.get("launches_per_step"), None if k.get("hbm_frac") is None else round(k["hbm_frac"],3))
PY
BENCH_EXTRA="" bash scripts/gpu_lib_ab.sh dpp mfma2 2>&1 | tee $OUT/ab_mfma_r04a.log
timeout 600 $ROOT/oracle/_ref/model_forward_shim --gpu-ordered > $OUT/shim_ordered_r04a.log 2>&1; cat $OUT/shim_ordered_r04a.log
GSPLAT_HIP_LIB=$ROOT/opensplat_amd/csrc/libgsplat_hip_stats.so timeout 300 python scripts/work_stats.py C2 > $OUT/work_stats_r04a_c2.json 2>> $OUT/bench_r04a.err; cat $OUT/work_stats_r04a_c2.json
# pixels per lane of the backward with the masks of round 3 and either reduction (the 338 / 378 / 500 us of DESIGN §4.1 predate the masks)
for px in 1 2; do for v in default dpp; do
  if [ "$v" = default ]; then unset GSPLAT_HIP_LIB; else export GSPLAT_HIP_LIB=$ROOT/opensplat_amd/csrc/libgsplat_hip_$v.so; fi
  GSPLAT_BWD_PX=$px python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('px$px $v', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernel_ms'].items()})" | tee -a $OUT/ab_px_r04a.log
done; done
unset GSPLAT_HIP_LIB
