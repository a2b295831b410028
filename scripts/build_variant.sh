#!/bin/bash
# Builds a variant of libgsplat_hip.so for A/B measurements:  scripts/build_variant.sh <name> <extra hipcc flags>
#   -> opensplat_amd/csrc/libgsplat_hip_<name>.so ; run with GSPLAT_HIP_LIB=<that path> python bench.py ...
set -e
NAME=$1; shift
cd "$(dirname "$0")/../opensplat_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-slp-vectorize \
    -Wno-unused-function "$@" gs_api.hip gs_project.hip gs_sh.hip gs_bin.hip gs_raster.hip gs_loss.hip gs_adam.hip \
    gs_densify.hip gs_fused.hip gs_compat.hip -o libgsplat_hip_$NAME.so
echo built libgsplat_hip_$NAME.so
