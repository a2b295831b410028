#!/bin/bash
# what the VALU counters read for SATURATED loops of each instruction class at the occupancy
# of the compositing kernels (4 - 5 waves per SIMD) — so that "VALU busy" has a measured ceiling.
#   plain run: cycles per wave-instruction per class (valu_rate's own event timing)
#   rocprofv3 --pmc pass: SQ_ACTIVE_INST_VALU, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_INSTS_VALU, GRBM_GUI_ACTIVE per kernel
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/valu_calib_$TAG
mkdir -p $OUT
cd $ROOT/scripts/ubench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate || exit 1
./valu_rate 4 5 8 > $OUT/valu_rate.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for w in 4 5; do
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_w$w -o pmc -- $ROOT/scripts/ubench/valu_rate $w > $OUT/pmc_w$w.log 2>&1
done
python3 $ROOT/scripts/summarize_valu_calib.py $TAG
