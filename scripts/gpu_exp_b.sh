#!/bin/bash
# round 6, session 3: the pipelined walk of backward_wave_q (default) against the round-5 walk (nopipe)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06s3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_deterministic.py tests/test_gpu_visibility_threshold.py tests/test_gpu_baseline_parity.py tests/test_gpu_parity_r03.py -x -q -m gpu 2>&1 | tail -5
REPS="1 2 3" bash scripts/gpu_lib_ab.sh nopipe 2>&1 | tee gpurun_out/r06s3/ab_pipe.log
BENCH_EXTRA="--config c3 --steps 20" REPS="1 2" bash scripts/gpu_lib_ab.sh nopipe 2>&1 | tee gpurun_out/r06s3/ab_pipe_c3.log
for c in C2 C3; do WAVE_TIMELINE_RAW=gpurun_out/r06s3/wlogp GSPLAT_HIP_LIB=$PWD/opensplat_amd/csrc/libgsplat_hip_wavelog.so timeout 400 python scripts/wave_timeline.py $c > gpurun_out/r06s3/wave_timeline_pipe_$c.json 2> gpurun_out/r06s3/wave_timeline_pipe_$c.err; done
