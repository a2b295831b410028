#!/bin/bash
# Builds oracle/_ref/model_forward_shim: tests/integration/model_forward_tu.cpp against a SCRATCH copy
# of the reference patched by integration/apply_hip_native.py.  Needs /root/reference (build
# container only); everything derived from the reference stays under oracle/_ref (git-ignored).
set -eu
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
REF=${REF:-/root/reference}
OUT=$ROOT/oracle/_ref/integration
CSRC=$ROOT/opensplat_amd/csrc
TORCH=$(python3 -c "import torch,os;print(os.path.dirname(torch.__file__))")
mkdir -p $OUT/overlay/gsplat $OUT/overlay/gsplat-cpu
python3 $ROOT/integration/apply_hip_native.py $REF --out $OUT/overlay
cp $REF/rasterizer/gsplat/config.h $OUT/overlay/gsplat/
cp $REF/rasterizer/gsplat-cpu/bindings.h $OUT/overlay/gsplat-cpu/
ABI=${GS_CXX11_ABI:-$(python3 -c "import torch;print(int(torch._C._GLIBCXX_USE_CXX11_ABI))")}
CXXFLAGS="-std=c++17 -O1 -w -D_GLIBCXX_USE_CXX11_ABI=$ABI -DUSE_HIP -DUSE_HIP_NATIVE -D__HIP_PLATFORM_AMD__=1 \
  -I$OUT/overlay -I$CSRC -I$TORCH/include -I$TORCH/include/torch/csrc/api/include -I/opt/rocm/include"
pids=""
for f in project_gaussians rasterize_gaussians spherical_harmonics; do
  g++ $CXXFLAGS -c $OUT/overlay/$f.cpp -o $OUT/$f.o & pids="$pids $!"
done
g++ $CXXFLAGS -c $ROOT/tests/integration/model_forward_tu.cpp -o $OUT/model_forward_tu.o & pids="$pids $!"
for p in $pids; do wait $p; done
# gsplat_cpu.cpp is untouched by the patch: the object oracle/Makefile built from it is reused
g++ -o $ROOT/oracle/_ref/model_forward_shim $OUT/model_forward_tu.o $OUT/project_gaussians.o \
  $OUT/rasterize_gaussians.o $OUT/spherical_harmonics.o $ROOT/oracle/_ref/gsplat_cpu.o \
  -Wl,--no-as-needed -L$TORCH/lib -ltorch -ltorch_cpu -lc10 -ltorch_hip -lc10_hip -L$CSRC -lgsplat_torch -lgsplat_hip \
  -Wl,--disable-new-dtags -Wl,-rpath,$CSRC -Wl,-rpath,$TORCH/lib
echo built $ROOT/oracle/_ref/model_forward_shim
