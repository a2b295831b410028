python -m pytest tests/test_gpu_block_masks.py tests/test_gpu_parity_r03.py tests/test_gpu_parity.py tests/test_gpu_deterministic.py -x -q 2>&1 | tail -5
B="python bench.py --no-cpu-baseline --steps 30"
for v in "" np "" np; do
  if [ -n "$v" ]; then export GSPLAT_HIP_LIB=$PWD/opensplat_amd/csrc/libgsplat_hip_$v.so; else unset GSPLAT_HIP_LIB; fi
  $B > gpurun_out/ab_${v:-def}.json 2>/dev/null
  $B --config c3 --steps 15 > gpurun_out/ab_${v:-def}_c3.json 2>/dev/null
  python - <<PY
import json
for f in ["", "_c3"]:
    d=json.loads(open("gpurun_out/ab_${v:-def}%s.json"%f).read().strip().splitlines()[-1])
    print("${v:-def}"+f, round(d["value"],1), "/s", round(d["ms_per_step"],4), "ms", {k:round(v,4) for k,v in d["kernel_ms"].items()}, {k:round(v,3) for k,v in d["stage_ms"].items()})
PY
done
unset GSPLAT_HIP_LIB
cd /tmp && export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_q2 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; rm -f gpurun_out/prof_q2/t_kernel_trace.csv
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/prof_q2/t_kernel_stats.csv")):
    n=r["Name"].split("(")[0].replace("void ","")
    print("  %-50s %4s %9.1f us  %5.1f%%"%(n[:50], r["Calls"], float(r["AverageNs"])/1000, float(r["Percentage"])))
PY
