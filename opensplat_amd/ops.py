"""Python view of the C++ autograd operators (opensplat_amd/csrc/torch_ops.cpp).

Names, argument order and meaning follow OpenSplat's operators:
  project_gaussians    <- ProjectGaussians::apply    (project_gaussians.hpp:12-30)
  rasterize_gaussians  <- RasterizeGaussians::apply  (rasterize_gaussians.hpp:23-37)
  spherical_harmonics  <- SphericalHarmonics::apply  (spherical_harmonics.hpp:15-22)
Inputs must be float32/int32 tensors on the GPU; anything else raises (c10::Error -> RuntimeError),
as the reference's CHECK_INPUT does (rasterizer/gsplat/bindings.h:14-19).
"""
from __future__ import annotations

import os

import torch

from . import _build

_LIB = _build.TORCH_LIB
if not os.path.exists(_LIB) or not os.path.exists(_build.HIP_LIB):
    raise ImportError(
        "opensplat_amd native libraries are not built (%s). Run `python -m opensplat_amd._build`; "
        "there is no CPU/PyTorch fallback." % _LIB)
torch.ops.load_library(_LIB)
_ops = torch.ops.opensplat_amd
_abi = list(_ops.abi_versions())
if _abi[0] != _abi[1]:   # (arguments would be shifted: refuse before the first call)
    raise ImportError("libgsplat_hip.so has ABI version %d, libgsplat_torch.so was built against %d: rebuild both "
                      "(python -m opensplat_amd._build --force)" % (_abi[0], _abi[1]))

BLOCK_X = BLOCK_Y = 16  # rasterizer/gsplat/config.h:1-2


def project_gaussians(means, scales, glob_scale, quats, viewmat, projmat, fx, fy, cx, cy,
                      img_height, img_width, tile_bounds=None, clip_thresh=0.01):
    """-> [xys, depths, radii, conics, num_tiles_hit, cov3d, cov2d] (first six = the reference's)."""
    return _ops.project_gaussians(means, scales, float(glob_scale), quats, viewmat, projmat,
                                  float(fx), float(fy), float(cx), float(cy), int(img_height),
                                  int(img_width), float(clip_thresh))


def rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height,
                        img_width, background, cov2d=None):
    """-> image [H, W, 3].  `cov2d` (7th output of project_gaussians) gives exact gsplat-cpu pixel-rectangle
    semantics; when it is omitted (the reference's ten-argument form) it is recovered from the storage
    project_gaussians shares between conics and cov2d — only a conics tensor that is not that operator's
    untouched output has its rectangle re-derived from the conic (cov2d_channel_counters)."""
    return _ops.rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacity,
                                    int(img_height), int(img_width), background, cov2d)


def spherical_harmonics(degrees_to_use, viewdirs, coeffs):
    """-> colors [N, 3] for coeffs [N, K, 3], K in {1, 4, 9, 16, 25}."""
    return _ops.spherical_harmonics(int(degrees_to_use), viewdirs, coeffs)


def set_fast_exp(enabled: bool) -> None:
    """Switch the compositing kernels to the hardware exp (not bit-compatible with gsplat-cpu)."""
    _ops.set_fast_exp(bool(enabled))


def set_segmented_backward(enabled: bool) -> None:
    """splat_render on frames of few tiles: checkpointed forward + segmented compositing backward (default on;
    scheduling only — gsplatSetSegmentedBackward)."""
    _ops.set_segmented_backward(bool(enabled))


def binning_reset() -> None:
    """Forget the id-list capacities of the speculative binning (gsplatResetBinningState)."""
    _ops.binning_reset()


def binning_counters():
    """-> (binning calls, forwards repeated because the id-list capacity was too small) since the
    last binning_reset()."""
    c = _ops.binning_counters()
    return int(c[0]), int(c[1])


def cov2d_channel_counters(reset: bool = False):
    """-> (hits, misses) of the reference-signature (ten-argument) rasterize calls: a hit found the
    frame's cov2d behind project_gaussians' own conics tensor, a miss inverted the conic."""
    c = _ops.cov2d_channel_counters(bool(reset))
    return int(c[0]), int(c[1])


def binning_capacity(device: int, img_width: int, img_height: int) -> int:
    """Id-list capacity currently held for (device, image size): the running maximum of 1.125 M + 1024."""
    return int(_ops.binning_capacity(int(device), int(img_width), int(img_height)))


def deg_from_sh(num_bases: int) -> int:  # spherical_harmonics.cpp:3-16
    return {1: 0, 4: 1, 9: 2, 16: 3}.get(int(num_bases), 4)


_C0 = 0.28209479177387814


def rgb2sh(rgb):  # spherical_harmonics.cpp:20-23
    return (rgb - 0.5) / _C0


def sh2rgb(sh):  # spherical_harmonics.cpp:25-28
    return torch.clamp(sh * _C0 + 0.5, 0.0, 1.0)


def splat_render(means, log_scales, quats, opacity_logits, features_dc, features_rest, viewmat,
                 projmat, cam_pos, fx, fy, cx, cy, img_height, img_width, degrees_to_use, background,
                 xys_grad_out=None):
    """Model::forward's render chain as one autograd node: raw parameters in,
    -> [rgb (clamped to <= 1), xys (detached), radii]; d loss / d xys is written to xys_grad_out."""
    return _ops.splat_render(means, log_scales, quats, opacity_logits, features_dc, features_rest,
                             viewmat, projmat, cam_pos, fx, fy, cx, cy, img_height, img_width,
                             degrees_to_use, background, xys_grad_out)


def camera_batch_step(means, log_scales, quats, opacity_logits, features_dc, features_rest, viewmats, projmats,
                      cam_pos, fx, fy, cx, cy, img_height, img_width, degrees_to_use, background, v_out,
                      deterministic=False, serial=False):
    """gsplat_ops.hpp CameraBatch::forwardBackward with fixed cotangents v_out [c,H,W,3] (test face): c cameras
    over the same raw parameters with two of them in flight -> [v_means, v_log_scales, v_quats, v_opacity_logits,
    v_features_dc, v_features_rest, rgb [c,H,W,3]]; the gradients are the sums over the cameras, in camera order."""
    return _ops.camera_batch_step(means, log_scales, quats, opacity_logits, features_dc, features_rest, viewmats,
                                  projmats, cam_pos, fx, fy, cx, cy, img_height, img_width, degrees_to_use,
                                  background, v_out, deterministic, serial)


def main_loss(rgb, gt, ssim_weight=0.2):
    """Model::mainLoss (model.cpp:780-784) as one autograd node: (1 - w) * L1 + w * (1 - SSIM) with the
    reference's 11x11 window; returns a 0-dim tensor, differentiable w.r.t. rgb (row f2)."""
    return _ops.main_loss(rgb, gt, float(ssim_weight))


def adam_step(params, grads, exp_avg, exp_avg_sq, lrs, step):
    """Model::optimizersStep (model.cpp:236-243) for up to eight parameter groups in ONE launch;
    params / exp_avg / exp_avg_sq are updated in place; `step` is 1-based."""
    _ops.adam_step(list(params), list(grads), list(exp_avg), list(exp_avg_sq),
                   [float(x) for x in lrs], int(step))


def densify_stats(xys_grad, radii, last_height, last_width, stats=None):
    """Model::afterTrain's per-iteration statistics (model.cpp:317-337).  stats = (xysGradNorm,
    visCounts, max2DSize) from the previous call, or None right after a refinement; returns them."""
    empty = torch.empty(0, device=radii.device)
    a, b, c = stats if stats is not None else (empty, empty.clone(), empty.clone())
    return tuple(_ops.densify_stats(xys_grad, radii, int(last_height), int(last_width), a, b, c))


def densify(params, exp_avg, exp_avg_sq, stats, last_width, last_height, densify_grad_thresh=0.0002,
            densify_size_thresh=0.01, check_screen_size=True, split_screen_size=0.05, cull_huge=True):
    """One refinement (model.cpp:345-458 + optimiser-state surgery).  params / exp_avg / exp_avg_sq:
    lists [means, scales(log), quats, opacities(logit, [N,1]), featuresDc, featuresRest] (moment
    lists may be empty).  -> (params, exp_avg, exp_avg_sq, dict(n_splits, n_dups, added, culled))."""
    out = _ops.densify(list(params), list(exp_avg or []), list(exp_avg_sq or []), stats[0], stats[1],
                       stats[2], int(last_width), int(last_height), float(densify_grad_thresh),
                       float(densify_size_thresh), bool(check_screen_size), float(split_screen_size),
                       bool(cull_huge))
    counts = dict(zip(["n_splits", "n_dups", "added", "culled"], [int(x) for x in out[-1]]))
    body = out[:-1]
    if len(body) == 18:
        return body[:6], body[6:12], body[12:18], counts
    return body[:6], None, None, counts
