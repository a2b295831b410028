#!/bin/bash
# Builds oracle/_ref/model_fused_shim: the reference's model.cpp patched by
# `integration/apply_hip_native.py --fused`, compiled for the GPU configuration (-DUSE_HIP
# -DUSE_HIP_NATIVE -DUSE_HIP_NATIVE_FUSED) against declaration-only stand-ins for OpenCV / nanoflann /
# json (oracle/stubs), + tests/integration/model_fused_tu.cpp as the training loop.  Needs
# /root/reference and the objects oracle/Makefile and build_model_forward_shim.sh leave in oracle/_ref.
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
REF=${REF:-/root/reference}
OUT=$ROOT/oracle/_ref/integration
CSRC=$ROOT/opensplat_amd/csrc
TORCH=$(python3 -c "import torch,os;print(os.path.dirname(torch.__file__))")
ABI=${GS_CXX11_ABI:-$(python3 -c "import torch;print(int(torch._C._GLIBCXX_USE_CXX11_ABI))")}
REFALPHA=${GS_FUSED_REFERENCE_ALPHA_RESET:+-DGS_FUSED_REFERENCE_ALPHA_RESET}
SUFFIX=${GS_FUSED_REFERENCE_ALPHA_RESET:+_refalpha}
mkdir -p $OUT/overlay_fused
# a scratch copy of the headers (model.hpp includes its siblings by relative path), then the patch
cp $REF/*.hpp $OUT/overlay_fused/
python3 $ROOT/integration/apply_hip_native.py $REF --out $OUT/overlay_fused --fused
CXXFLAGS="-std=c++17 -O1 -w -D_GLIBCXX_USE_CXX11_ABI=$ABI -DUSE_HIP -DUSE_HIP_NATIVE -DUSE_HIP_NATIVE_FUSED $REFALPHA \
  -D__HIP_PLATFORM_AMD__=1 -I$OUT/overlay_fused -I$ROOT/oracle/stubs -I$CSRC -I$REF/rasterizer \
  -I$TORCH/include -I$TORCH/include/torch/csrc/api/include -I/opt/rocm/include"
g++ $CXXFLAGS -c $OUT/overlay_fused/model.cpp -o $OUT/model_fused$SUFFIX.o &
g++ $CXXFLAGS -c $ROOT/tests/integration/model_fused_tu.cpp -o $OUT/model_fused_tu$SUFFIX.o &
wait
# the operator files (patched, GPU parts compiled out) come from build_model_forward_shim.sh; ssim,
# optim_scheduler, tensor_math and gsplat_cpu from oracle/Makefile (they have no GPU variant)
g++ -o $ROOT/oracle/_ref/model_fused_shim$SUFFIX $OUT/model_fused_tu$SUFFIX.o $OUT/model_fused$SUFFIX.o \
  $OUT/project_gaussians.o $OUT/rasterize_gaussians.o $OUT/spherical_harmonics.o \
  $ROOT/oracle/_ref/ssim.o $ROOT/oracle/_ref/optim_scheduler.o $ROOT/oracle/_ref/tensor_math.o $ROOT/oracle/_ref/gsplat_cpu.o \
  -Wl,--no-as-needed -L$TORCH/lib -ltorch -ltorch_cpu -lc10 -ltorch_hip -lc10_hip -L$CSRC -lgsplat_torch -lgsplat_hip \
  -Wl,--disable-new-dtags -Wl,-rpath,$CSRC -Wl,-rpath,$TORCH/lib
echo built $ROOT/oracle/_ref/model_fused_shim$SUFFIX
