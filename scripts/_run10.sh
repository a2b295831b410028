timeout 1500 python -m pytest tests/test_gpu_ops_and_edges.py tests/test_gpu_parity.py tests/test_gpu_block_masks.py tests/test_gpu_baseline_parity.py -x -q 2>&1 | tail -5
B="python bench.py --no-cpu-baseline --steps 30"
$B > gpurun_out/r10.json 2>/dev/null
$B --config c3 --steps 15 > gpurun_out/r10_c3.json 2>/dev/null
python - <<PY
import json
for f in ["", "_c3"]:
    d=json.loads(open("gpurun_out/r10%s.json"%f).read().strip().splitlines()[-1])
    print("r10"+f, round(d["value"],1), "/s", round(d["ms_per_step"],4), "ms", {k:round(v,4) for k,v in d["kernel_ms"].items()}, {k:round(v,3) for k,v in d["stage_ms"].items()})
PY
cd /tmp && export TMPDIR=/tmp
for c in c2 c3; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_q4$c -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --config $c > /dev/null 2>&1
rm -f $GRAFT_REPO_ROOT/gpurun_out/prof_q4$c/t_kernel_trace.csv
python - <<PY
import csv
for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/prof_q4$c/t_kernel_stats.csv")):
    n=r["Name"].split("(")[0].replace("void ","")
    if float(r["Percentage"])>0.3: print("  %-50s %4s %9.1f us  %5.1f%%"%(n[:50], r["Calls"], float(r["AverageNs"])/1000, float(r["Percentage"])))
PY
done
