#!/bin/bash
# the extended random sweep of the compositing kernels (tests/extended_sweep.py) on the GPU box
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
( time timeout ${SWEEP_TIMEOUT:-800} python tests/extended_sweep.py ${SWEEP_RANGE:-15 2015} ) > gpurun_out/extended_sweep_${SWEEP_TAG:-r05}.log 2>&1
tail -5 gpurun_out/extended_sweep_${SWEEP_TAG:-r05}.log
