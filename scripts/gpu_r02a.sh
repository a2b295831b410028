#!/bin/bash
# Round-2 GPU call A: the whole -m gpu suite (incl. the BASELINE-size parity tests), bench lines,
# the moving-camera measurement and the counter calibration on gather / atomic patterns.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest_r02a.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest_r02a.log
tail -5 $OUT/pytest_r02a.log
timeout 300 python bench.py > $OUT/bench_r02a.json 2> $OUT/bench_r02a.err
timeout 300 python bench.py --config c4-sequence --steps 48 --warmup 0 --no-cpu-baseline > $OUT/bench_r02a_c4seq.json 2> $OUT/bench_r02a_c4seq.err
timeout 300 python bench.py --config c4-sequence --steps 48 --warmup 8 --no-cpu-baseline > $OUT/bench_r02a_c4seq_warm.json 2>> $OUT/bench_r02a_c4seq.err
cd /tmp && export TMPDIR=/tmp
G=$ROOT/scripts/ubench/gather_calib
timeout 300 $G > $OUT/calib_plain.jsonl 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -o fetch -- $G > $OUT/calib_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calib_write -o write -- $G > $OUT/calib_write.log 2>&1
ls $OUT/calib_fetch $OUT/calib_write
