#!/usr/bin/env python3
"""Patches an OpenSplat checkout so that its OWN call sites (model.cpp:114-222,
simple_trainer.cpp:152-192) build against the MI355X-native operators of libgsplat_torch.so while the
CPU classes (ProjectGaussiansCPU / RasterizeGaussiansCPU / SphericalHarmonicsCPU) stay exactly
where they are — both are used in the same translation unit (model.cpp:123-135,182,195-206).

    python integration/apply_hip_native.py /path/to/OpenSplat [--out DIR]

Edits (all under a new macro USE_HIP_NATIVE, so every other configuration is untouched; nothing is
copied out of the checkout, the seven files are edited in place or into --out):

  project_gaussians.hpp / rasterize_gaussians.hpp / spherical_harmonics.hpp
      the `#if defined(USE_HIP) || defined(USE_CUDA) || defined(USE_MPS)` block that DECLARES the GPU
      class (and binAndSortGaussians) becomes
          #ifdef USE_HIP_NATIVE
          #include "gsplat_ops.hpp"        // this repo: same class names / argument order
          #else  <the original block>  #endif
      the CPU class declarations below it are not touched.
  project_gaussians.cpp / rasterize_gaussians.cpp / spherical_harmonics.cpp
      the same guard around the GPU class DEFINITIONS gets `&& !defined(USE_HIP_NATIVE)` (they call
      rasterizer/gsplat's CUDA launchers); the CPU definitions below stay.
  gsplat.hpp
      `#include <gsplat/bindings.h>` (pulls forward.cuh / glm) is skipped under USE_HIP_NATIVE.

CMake (print with --cmake): a GPU_RUNTIME value HIP_NATIVE that defines USE_HIP USE_HIP_NATIVE, adds
this repo's opensplat_amd/csrc to the include path and links libgsplat_torch.so + libgsplat_hip.so
instead of building rasterizer/gsplat.

--fused additionally patches model.cpp (macro USE_HIP_NATIVE_FUSED) so that OpenSplat's own `Model`
runs on the fused operators: five insertions, each a guarded early return into
opensplat_amd/csrc/model_fused.inl —
    #include "model_fused.inl"                                  after model.cpp's own includes
    Model::forward        -> gs_fused::render(...)              in front of the torch::cat of the SH
                                                                coefficients (model.cpp:114)
    Model::optimizersStep -> gs_fused::optimizers_step(*this)   model.cpp:236
    Model::afterTrain     -> gs_fused::after_train(*this, step) model.cpp:311
    Model::mainLoss       -> ::mainLoss(rgb, gt, ssimWeight)    model.cpp:780
The CPU device keeps every original statement.

tests/test_integration_build.py applies this script to a scratch copy of the reference and builds +
runs a translation unit shaped like Model::forward against the result; tests/test_gpu_model_fused.py
builds the patched model.cpp itself and trains with it on the GPU.
"""
import argparse
import os
import re
import shutil
import sys

GUARD = "#if defined(USE_HIP) || defined(USE_CUDA) || defined(USE_MPS)"
HEADERS = ["project_gaussians.hpp", "rasterize_gaussians.hpp", "spherical_harmonics.hpp"]
SOURCES = ["project_gaussians.cpp", "rasterize_gaussians.cpp", "spherical_harmonics.cpp"]

CMAKE_SNIPPET = r'''
# ---- OpenSplat CMakeLists.txt: MI355X-native rasterizer (GPU_RUNTIME=HIP_NATIVE) -----------------
elseif(GPU_RUNTIME STREQUAL "HIP_NATIVE")
    set(GSPLAT_AMD_DIR "" CACHE PATH "opensplat_amd/csrc of the MI355X rasterizer repo")
    find_library(GSPLAT_HIP   gsplat_hip   HINTS ${GSPLAT_AMD_DIR} REQUIRED)   # hand-written gfx950 kernels
    find_library(GSPLAT_TORCH gsplat_torch HINTS ${GSPLAT_AMD_DIR} REQUIRED)   # libtorch operators
    add_library(gsplat_cpu rasterizer/gsplat-cpu/gsplat_cpu.cpp)               # unchanged
    # rasterizer/gsplat is NOT built.  libtorch first: with a pip-wheel libtorch-ROCm (which bundles its own
    # libamdhip64.so / librccl.so) the wheel's runtime must be mapped before libgsplat_hip asks for
    # "libamdhip64.so.7", or the process ends up with two HIP runtimes
    set(GSPLAT_LIBS gsplat_cpu ${TORCH_LIBRARIES} ${GSPLAT_TORCH} ${GSPLAT_HIP})
    add_compile_definitions(USE_HIP USE_HIP_NATIVE __HIP_PLATFORM_AMD__)
    include_directories(${GSPLAT_AMD_DIR})                                     # gsplat_ops.hpp
'''


def patch_header(text: str, name: str) -> str:
    if "USE_HIP_NATIVE" in text:
        return text
    i = text.find(GUARD)
    if i < 0:
        raise SystemExit("%s: GPU-class guard not found (different OpenSplat version?)" % name)
    j = text.find("#endif", i)
    if j < 0:
        raise SystemExit("%s: unterminated guard" % name)
    j += len("#endif")
    block = text[i:j]
    new = ("#ifdef USE_HIP_NATIVE\n"
           "// MI355X-native GPU operators (libgsplat_torch.so): same class names, argument order and\n"
           "// return order as the declarations in the #else branch\n"
           "#include \"gsplat_ops.hpp\"\n"
           "#else\n" + block + "\n#endif")
    return text[:i] + new + text[j:]


def patch_source(text: str, name: str) -> str:
    if "USE_HIP_NATIVE" in text:
        return text
    if GUARD not in text:
        raise SystemExit("%s: GPU-definition guard not found" % name)
    return text.replace(GUARD, "#if (defined(USE_HIP) || defined(USE_CUDA) || defined(USE_MPS)) && "
                               "!defined(USE_HIP_NATIVE)", 1)


def patch_gsplat_hpp(text: str) -> str:
    if "USE_HIP_NATIVE" in text:
        return text
    pat = re.compile(r"#if defined\(USE_HIP\) \|\| defined\(USE_CUDA\)\s*\n(\s*#include <gsplat/bindings.h>)")
    if not pat.search(text):
        raise SystemExit("gsplat.hpp: bindings.h include not found")
    return pat.sub(r"#if (defined(USE_HIP) || defined(USE_CUDA)) && !defined(USE_HIP_NATIVE)\n\1", text, 1)


FUSED_HUNKS = [
    # (anchor text that must occur exactly once, text inserted IN FRONT of the anchor's line)
    ('namespace fs = std::filesystem;',
     '#ifdef USE_HIP_NATIVE_FUSED\n'
     '// MI355X-native fused operators behind Model\'s own call sites (opensplat_amd/csrc)\n'
     '#include "model_fused.inl"\n'
     '#endif\n\n'),
    # in front of the camera matrices: on a GPU device the reference builds them ON the device
    # (torch::eye(4, device) + two index_put_ of host tensors, projectionMatrix(..., device)): three
    # synchronising host-to-device copies per iteration; the fused path keeps them on the host and hands
    # them to the kernels by value
    ('    torch::Tensor viewMat = torch::eye(4, device);',
     '#ifdef USE_HIP_NATIVE_FUSED\n'
     '    if (device != torch::kCPU){\n'
     '        const float fovX_ = 2.0f * std::atan(width / (2.0f * fx)), fovY_ = 2.0f * std::atan(height / (2.0f * fy));\n'
     '        return gs_fused::render(*this, Rinv, Tinv, T, projectionMatrix(0.001f, 1000.0f, fovX_, fovY_, torch::kCPU),\n'
     '                                fx, fy, cx, cy, height, width, step);\n'
     '    }\n'
     '#endif\n'),
    ('  meansOpt->step();',
     '#ifdef USE_HIP_NATIVE_FUSED\n'
     '  if (device != torch::kCPU){ gs_fused::optimizers_step(*this); return; }\n'
     '#endif\n'),
    ('    // When radii.sum() == 0',
     '#ifdef USE_HIP_NATIVE_FUSED\n'
     '    if (device != torch::kCPU){ gs_fused::after_train(*this, step); return; }\n'
     '#endif\n'),
    ('    torch::Tensor ssimLoss = 1.0f - ssim.eval(rgb, gt);',
     '#ifdef USE_HIP_NATIVE_FUSED\n'
     '    if (rgb.is_cuda()) return ::mainLoss(rgb, gt, ssimWeight);\n'
     '#endif\n'),
]


def patch_model_cpp(text: str) -> str:
    if "USE_HIP_NATIVE_FUSED" in text:
        return text
    for anchor, insert in FUSED_HUNKS:
        if text.count(anchor) != 1:
            raise SystemExit("model.cpp: anchor %r found %d times (different OpenSplat version?)"
                             % (anchor, text.count(anchor)))
        i = text.index(anchor)
        line_start = text.rfind("\n", 0, i) + 1
        text = text[:line_start] + insert + text[line_start:]
    return text


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("checkout", nargs="?", help="root of the OpenSplat source tree")
    ap.add_argument("--out", help="write the patched files here instead of editing in place")
    ap.add_argument("--cmake", action="store_true", help="print the CMakeLists.txt snippet and exit")
    ap.add_argument("--fused", action="store_true",
                    help="also patch model.cpp onto SplatRender / MainLoss / the fused Adam step / densify "
                         "(NB: by default the alpha reset keeps the opacities trainable — the evident intent — "
                         "where the reference freezes them until the next refinement: trajectories differ from "
                         "the first reset on; build with -DGS_FUSED_REFERENCE_ALPHA_RESET for the reference's "
                         "behaviour, INTEGRATION.md §7)")
    a = ap.parse_args()
    if a.cmake or not a.checkout:
        print(CMAKE_SNIPPET)
        return
    out = a.out or a.checkout
    os.makedirs(out, exist_ok=True)
    jobs = [(h, patch_header) for h in HEADERS] + [(s, patch_source) for s in SOURCES]
    for name, fn in jobs:
        text = open(os.path.join(a.checkout, name)).read()
        open(os.path.join(out, name), "w").write(fn(text, name))
    text = open(os.path.join(a.checkout, "gsplat.hpp")).read()
    open(os.path.join(out, "gsplat.hpp"), "w").write(patch_gsplat_hpp(text))
    if a.fused:
        text = open(os.path.join(a.checkout, "model.cpp")).read()
        open(os.path.join(out, "model.cpp"), "w").write(patch_model_cpp(text))
    if a.out:   # the untouched headers the patched ones include
        shutil.copy(os.path.join(a.checkout, "tile_bounds.hpp"), out)
    print("patched: %s" % ", ".join(HEADERS + SOURCES + ["gsplat.hpp"] + (["model.cpp"] if a.fused else [])),
          file=sys.stderr)


if __name__ == "__main__":
    main()
