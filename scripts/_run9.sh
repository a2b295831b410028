B="python bench.py --no-cpu-baseline --steps 30"
for px in 4 2 1 4 2; do
  GSPLAT_BWD_PX=$px $B > gpurun_out/px$px.json 2>/dev/null
  GSPLAT_BWD_PX=$px $B --config c3 --steps 12 > gpurun_out/px${px}_c3.json 2>/dev/null
  python - <<PY
import json
for f in ["", "_c3"]:
    d=json.loads(open("gpurun_out/px$px%s.json"%f).read().strip().splitlines()[-1])
    print("px$px"+f, round(d["value"],1), "/s", round(d["ms_per_step"],4), "ms", {k:round(v,4) for k,v in d["kernel_ms"].items()})
PY
done
