"""opensplat_amd — MI355X (gfx950) native differentiable Gaussian-splat rasterizer.

A drop-in for the hot path of pierotofy/OpenSplat behind its own operator surface
(ProjectGaussians / RasterizeGaussians / SphericalHarmonics), built as:

  include/gsplat_hip.h                 the C ABI (extern "C", raw device pointers + stream)
  opensplat_amd/csrc/*.hip             hand-written HIP kernels for gfx950 -> libgsplat_hip.so
  opensplat_amd/csrc/torch_ops.cpp     C++/libtorch autograd operators     -> libgsplat_torch.so
  opensplat_amd/ops.py                 Python view of the same operators (torch.ops.opensplat_amd)
  opensplat_amd/cabi.py                ctypes binding of the C ABI on torch tensors (tests, bench)

There is no CPU or PyTorch fallback: importing `opensplat_amd.ops` / `opensplat_amd.cabi` raises
if the native libraries have not been built (`python -m opensplat_amd._build`).
"""

__version__ = "0.1.0"
