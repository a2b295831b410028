#!/bin/bash
# kernel averages (rocprofv3 --kernel-trace --stats) of the default bench per library variant
#   usage: gpu_ktrace.sh suffix1 suffix2 ...   ("default" = the in-tree library)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then unset GSPLAT_HIP_LIB; else export GSPLAT_HIP_LIB=$ROOT/opensplat_amd/csrc/libgsplat_hip_$v.so; fi
  rm -rf /tmp/kt_$v
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o trace -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-} > /dev/null 2>&1
  f=$(find /tmp/kt_$v -name "*kernel_stats.csv" | head -1)
  python - <<PY
import csv
print("== $v")
for r in list(csv.DictReader(open("$f")))[:10]:
    print("  ", r['Name'][:46].ljust(48), r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
done
