#!/bin/bash
# round 6, session 3: claim tag as a byte (20 waves per CU) vs 32-bit (19), four waves per SIMD, per-wave timeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06s3
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deterministic.py tests/test_gpu_visibility_threshold.py -x -q -m gpu 2>&1 | tail -3
REPS="1 2 3" bash scripts/gpu_lib_ab.sh tag32 w4 2>&1 | tee gpurun_out/r06s3/ab_tag8.log
BENCH_EXTRA="--config c3 --steps 20" REPS="1 2" bash scripts/gpu_lib_ab.sh tag32 2>&1 | tee gpurun_out/r06s3/ab_tag8_c3.log
GSPLAT_HIP_LIB=$PWD/opensplat_amd/csrc/libgsplat_hip_stats.so timeout 300 python scripts/wave_timeline.py C2 > gpurun_out/r06s3/wave_timeline_c2.json 2> gpurun_out/r06s3/wave_timeline_c2.err
tail -3 gpurun_out/r06s3/wave_timeline_c2.err
