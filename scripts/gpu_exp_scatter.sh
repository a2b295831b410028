#!/bin/bash
# What k_scatter_scan's time hangs on: library variants -DGS_EXP_SCATTER=1..4 (scripts/build_variant.sh exps<n>) timed by
# rocprofv3 --kernel-trace --stats under scripts/exp_bin_only.py (binning only: nothing reads the wrong lists).
#   scripts/gpu_exp_scatter.sh [configs...]   -> gpurun_out/exp_scatter/<config>_<variant>.{json,csv}, summary on stdout
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/exp_scatter
mkdir -p $OUT
VARIANTS=${VARIANTS:-default exps1 exps2 exps3 exps4}
for cfg in "${@:-c2}"; do
for v in $VARIANTS; do
  if [ "$v" = default ]; then unset GSPLAT_HIP_LIB; else export GSPLAT_HIP_LIB=$ROOT/opensplat_amd/csrc/libgsplat_hip_$v.so; fi
  D=/tmp/exp_scatter_${cfg}_$v; rm -rf $D
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o x -- python $ROOT/scripts/exp_bin_only.py --config $cfg --iters ${ITERS:-200} > $OUT/${cfg}_$v.json 2> $OUT/${cfg}_$v.err)
  S=$(find $D -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && cp $S $OUT/${cfg}_${v}_kernel_stats.csv
  echo "== $cfg $v $(tail -1 $OUT/${cfg}_$v.json)"
  [ -n "$S" ] && python - "$S" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"] if "Name" in r else r.get("Kernel", "")
    if "gs::" in n or "fill" in n.lower():
        print("   %-60s calls %6s avg %9.1f us" % (n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1000.0))
P
done
done
