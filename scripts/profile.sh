#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   kernel trace + stats of the default bench command, then PMC passes (FETCH_SIZE, WRITE_SIZE, SQ)
# Outputs land in gpurun_out/prof_<tag>/ ; scripts/summarize_profile.py turns them into
# profiles/<tag>_*.{md,json}.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# BENCH_ARGS: extra bench.py arguments (e.g. "--config c3") for profiles of the other configurations
BENCH="python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
if [ "${TRACE_ONLY:-0}" = 1 ]; then find $OUT -name "*.csv" | head; exit 0; fi   # (kernel stats only: no PMC passes)
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/sq -o sq -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/sq2 -o sq2 -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/sq2.log 2>&1
find $OUT -name "*.csv" | head -40
# keep the merge small: drop the raw per-dispatch traces of the long run except stats
du -sh $OUT
