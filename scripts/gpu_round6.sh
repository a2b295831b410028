#!/bin/bash
# Round 6 evidence run: scripts/gpu_round.sh (LEAN, C3 counters), then everything that is summarised ON the box
# (the raw rocprofv3 traces are too large to merge back) copied under gpurun_out/profiles_r06/.
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
LEAN=1 C3_PMC=1 bash scripts/gpu_round.sh $TAG > gpurun_out/round_$TAG.log 2>&1
timeout 600 python scripts/bench_train_step.py > gpurun_out/f2_train_step_$TAG.json 2>> gpurun_out/bench_$TAG.err
# per-wave timeline of the two compositing kernels (scripts/build_variant.sh wavelog -DGS_WAVELOG) and a long random sweep
for c in C2 C3; do
  lc=$(echo $c | tr A-Z a-z)
  GSPLAT_HIP_LIB=$ROOT/opensplat_amd/csrc/libgsplat_hip_wavelog.so timeout 300 python scripts/wave_timeline.py $c > gpurun_out/wave_timeline_${TAG}_$lc.json 2>> gpurun_out/bench_$TAG.err
done
SWEEP_TAG=$TAG SWEEP_RANGE="${SWEEP_RANGE:-15 6015}" SWEEP_TIMEOUT=1500 bash scripts/gpu_sweep.sh > /dev/null 2>&1
for c in c2 c3; do cp gpurun_out/work_stats_${TAG}_$c.json profiles/work_stats_${TAG}_$c.json 2>/dev/null; done
python scripts/summarize_profile.py $TAG > /dev/null 2>&1
python scripts/summarize_profile.py ${TAG}_c3 > /dev/null 2>&1
python scripts/issue_roofline.py > gpurun_out/issue_roofline_$TAG.log 2>&1
mkdir -p gpurun_out/profiles_$TAG
cp gpurun_out/wave_timeline_${TAG}_c2.json gpurun_out/wave_timeline_${TAG}_c3.json gpurun_out/extended_sweep_${TAG}.log gpurun_out/profiles_$TAG/ 2>/dev/null
cp profiles/${TAG}_* profiles/kernels.json profiles/kernels_c3.json profiles/traffic.json profiles/traffic_c3.json \
   profiles/issue_roofline_r06.json gpurun_out/profiles_$TAG/ 2>/dev/null
# a last default bench line WITH the fresh replayed files in place (what the driver's run will print)
timeout 300 python bench.py > gpurun_out/bench_${TAG}_final.json 2>> gpurun_out/bench_$TAG.err
timeout 300 python bench.py --no-cpu-baseline --config c3 --steps 20 > gpurun_out/bench_${TAG}_final_c3.json 2>> gpurun_out/bench_$TAG.err
find gpurun_out -name "*_kernel_trace.csv" -size +2M -delete; find gpurun_out -name "*counter_collection.csv" -size +4M -delete
tail -30 gpurun_out/round_$TAG.log; cat gpurun_out/issue_roofline_$TAG.log; du -sh gpurun_out
