"""In-tree build of the two native libraries (no setuptools, no JIT cache):

  csrc/libgsplat_hip.so    HIP kernels + C ABI (include/gsplat_hip.h), hipcc --offload-arch=gfx950
  csrc/libgsplat_torch.so  libtorch autograd operators on top of the C ABI, g++

hipcc cross-compiles gfx950 without a GPU.  Both are rebuilt only when a source is newer than the
library.  The built .so files are git-ignored but travel with the gpurun snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
INCLUDE = os.path.join(ROOT, "include")

HIP_SOURCES = ["gs_api.hip", "gs_project.hip", "gs_sh.hip", "gs_bin.hip", "gs_raster.hip",
               "gs_loss.hip", "gs_adam.hip", "gs_densify.hip", "gs_fused.hip", "gs_compat.hip"]
HIP_HEADERS = ["gs_device.h", "gs_gaussian.h", os.path.join(INCLUDE, "gsplat_hip.h"),
               os.path.join(INCLUDE, "gsplat_train.h"), os.path.join(INCLUDE, "gsplat_densify.h"),
               os.path.join(INCLUDE, "gsplat_compat.h")]
HIP_LIB = os.path.join(CSRC, "libgsplat_hip.so")
TORCH_LIB = os.path.join(CSRC, "libgsplat_torch.so")

# -ffp-contract=off: the compositing arithmetic must round like the CPU reference (no implicit FMA)
# -fno-slp-vectorize: the SLP vectoriser pairs scalar fp32 operations of the compositing loops into
# v_pk_* instructions plus the v_mov shuffles that line their operands up; a packed operation issues
# in ~5 cycles against 2 x 2.9 for the plain ones, the moves eat the difference (measured: backward
# 386 -> 381 us, k_gaussian_backward 71 -> 67 us at C2, profiles/bench_va_*.json)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def _stale(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd: list[str]) -> None:
    print("[opensplat_amd build]", " ".join(cmd), file=sys.stderr, flush=True)
    subprocess.check_call(cmd, cwd=CSRC)


def build_hip(force: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HIP_HEADERS]
    if force or _stale(HIP_LIB, deps):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        _run([hipcc] + HIPCC_FLAGS + HIP_SOURCES + ["-o", HIP_LIB])
    return HIP_LIB


def build_torch(force: bool = False) -> str:
    import torch

    tdir = os.path.dirname(torch.__file__)
    src = os.path.join(CSRC, "torch_ops.cpp")
    launcher = os.path.join(CSRC, "bindings_hip_native.cpp")   # the eight *_tensor launchers
    deps = [src, launcher, os.path.join(CSRC, "bindings_hip_native.h"), os.path.join(CSRC, "gsplat_ops.hpp"),
            os.path.join(INCLUDE, "gsplat_hip.h"), os.path.join(INCLUDE, "gsplat_compat.h"),
            os.path.join(INCLUDE, "gsplat_dist.h"), HIP_LIB, DIST_LIB]
    if force or _stale(TORCH_LIB, deps):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-w",
              "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
              "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
              "-I" + os.path.join(tdir, "include"),
              "-I" + os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
              "-I/opt/rocm/include",
              "torch_ops.cpp", "bindings_hip_native.cpp", "-o", TORCH_LIB,
              # libtorch's libraries FIRST: the loader walks NEEDED entries breadth-first, and the
              # ROCm runtime the wheel bundles (libamdhip64.so, librccl.so; SONAMEs .so.7 / .so.1)
              # must be mapped before libgsplat_hip/_dist ask for "libamdhip64.so.7" /
              # "librccl.so.1" -- a request matches an already-loaded SONAME, not the other way
              # round, and a process with two HIP runtimes or two RCCLs corrupts its heap at exit.
              "-Wl,--no-as-needed",
              "-L" + os.path.join(tdir, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip",
              "-lc10_hip", "-L" + CSRC, "-lgsplat_hip", "-lgsplat_dist", "-Wl,-rpath,$ORIGIN"])
    return TORCH_LIB


DIST_LIB = os.path.join(CSRC, "libgsplat_dist.so")


def build_dist(force: bool = False) -> str:
    """csrc/libgsplat_dist.so: the gradient exchange on RCCL behind include/gsplat_dist.h (g++;
    links the RCCL libtorch-ROCm ships, so that a process never holds two RCCLs)."""
    import torch

    tdir = os.path.dirname(torch.__file__)
    src = os.path.join(CSRC, "gs_dist.cpp")
    deps = [src, os.path.join(INCLUDE, "gsplat_dist.h"), os.path.join(INCLUDE, "gsplat_hip.h")]
    if force or _stale(DIST_LIB, deps):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1",
              "-I/opt/rocm/include", "gs_dist.cpp", "-o", DIST_LIB,
              "-L" + os.path.join(tdir, "lib"), "-lrccl", "-L/opt/rocm/lib", "-lamdhip64",
              "-Wl,-rpath," + os.path.join(tdir, "lib"), "-Wl,-rpath,/opt/rocm/lib"])
    return DIST_LIB


IMAGE_LIB = os.path.join(CSRC, "libgsplat_image.so")


def build_image(force: bool = False) -> str:
    """csrc/libgsplat_image.so: the baseline JPEG decoder of include/gsplat_image.h (plain host C)."""
    src = os.path.join(CSRC, "gs_image.c")
    if force or _stale(IMAGE_LIB, [src, os.path.join(INCLUDE, "gsplat_image.h")]):
        _run(["gcc", "-std=c11", "-O2", "-fPIC", "-shared", "-Wall", "gs_image.c", "-o", IMAGE_LIB])
    return IMAGE_LIB


EXAMPLE_SRC = os.path.join(ROOT, "examples", "simple_trainer_hip.cpp")
EXAMPLE_BIN = os.path.join(ROOT, "examples", "simple_trainer_hip")


def build_example(force: bool = False) -> str:
    """examples/simple_trainer_hip: a C++ caller of gsplat_ops.hpp (BASELINE config 1), linked
    against the two in-tree libraries only."""
    import torch

    tdir = os.path.dirname(torch.__file__)
    deps = [EXAMPLE_SRC, os.path.join(CSRC, "gsplat_ops.hpp"), TORCH_LIB]
    if force or _stale(EXAMPLE_BIN, deps):
        _run(["g++", "-std=c++17", "-O2", "-w",
              "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
              "-I" + CSRC, "-I" + os.path.join(tdir, "include"),
              "-I" + os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
              EXAMPLE_SRC, "-o", EXAMPLE_BIN,
              "-Wl,--no-as-needed",   # torch's libraries first and kept: see build_torch on load order
              "-L" + os.path.join(tdir, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip",
              "-lc10_hip", "-L" + CSRC, "-lgsplat_torch", "-lgsplat_hip", "-lgsplat_dist",
              "-Wl,--disable-new-dtags",   # RPATH, so that it also serves libgsplat_torch.so's deps
              "-Wl,-rpath," + CSRC, "-Wl,-rpath," + os.path.join(tdir, "lib")])
    return EXAMPLE_BIN


def build_all(force: bool = False) -> None:
    build_hip(force)
    build_image(force)
    build_dist(force)
    build_torch(force)
    build_example(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
