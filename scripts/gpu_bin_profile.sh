#!/bin/bash
# rocprofv3 rows of the strip binning beside the tile-level one (round 6): kernel trace + stats at C2 / C3 for
# GSPLAT_BIN=strips (and the fused variant), WRITE_SIZE / FETCH_SIZE / SQ passes at C2.
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
GSPLAT_BIN=strips bash scripts/profile.sh ${TAG}_strips > /dev/null 2>&1
GSPLAT_BIN=strips TRACE_ONLY=1 BENCH_ARGS="--config c3" bash scripts/profile.sh ${TAG}_strips_c3 > /dev/null 2>&1
GSPLAT_BIN=strips GSPLAT_STRIPS_FUSED=1 TRACE_ONLY=1 bash scripts/profile.sh ${TAG}_fused > /dev/null 2>&1
GSPLAT_BIN=strips GSPLAT_STRIPS_FUSED=1 TRACE_ONLY=1 BENCH_ARGS="--config c3" bash scripts/profile.sh ${TAG}_fused_c3 > /dev/null 2>&1
GSPLAT_BIN=tiles bash scripts/profile.sh ${TAG}_tiles > /dev/null 2>&1
GSPLAT_BIN=tiles TRACE_ONLY=1 BENCH_ARGS="--config c3" bash scripts/profile.sh ${TAG}_tiles_c3 > /dev/null 2>&1
for t in strips strips_c3 fused fused_c3 tiles tiles_c3; do python scripts/summarize_profile.py ${TAG}_$t > /dev/null 2>&1; done
ls profiles | grep ${TAG}_
# keep the merge small
find gpurun_out -name "*_kernel_trace.csv" -size +3M -delete; find gpurun_out -name "*counter_collection.csv" -size +8M -delete
du -sh gpurun_out
