"""Shared helpers for the parity tests: run a Scene through the HIP C ABI and through the oracle."""
from __future__ import annotations

import numpy as np


def to_dev(a, dtype=None):
    import torch

    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def hip_pipeline(s, flags=0, backward=True, use_cov2d=True, device_matrices=False):
    """Scene -> dict of numpy arrays, every stage through include/gsplat_hip.h via ctypes."""
    import torch

    from opensplat_amd import cabi

    cam = cabi.make_camera(s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy, s.W, s.H)
    means, scales, quats = to_dev(s.means), to_dev(s.scales), to_dev(s.quats)
    opac = to_dev(s.opacities.reshape(-1))
    vm_dev = to_dev(s.viewmat) if device_matrices else None
    pm_dev = to_dev(s.projmat) if device_matrices else None
    p = cabi.project_forward(cam, means, scales, quats, vm_dev, pm_dev)
    out = {k: v for k, v in p.items()}
    if s.sh_coeffs is not None:
        dirs, coeffs = to_dev(s.dirs), to_dev(s.sh_coeffs)
        sh_rgb = cabi.sh_forward(s.degrees_to_use, dirs, coeffs)
        colors = torch.clamp_min(sh_rgb + 0.5, 0.0)  # model.cpp:192 (caller-side torch op)
        out["sh_rgb"] = sh_rgb
    else:
        colors = to_dev(s.colors)
    out["colors"] = colors
    b = cabi.bin_and_sort(s.W, s.H, p["xys"], p["depths"], p["radii"], p["conics"], colors, opac,
                          p["cov2d"] if use_cov2d else None)
    out["binned"] = b
    f = cabi.rasterize_forward(s.W, s.H, b, s.background, flags)
    out.update(f)
    if backward and s.v_out is not None:
        v_out = to_dev(s.v_out)
        g = cabi.rasterize_backward(s.W, s.H, s.N, b, s.background, f["final_Ts"], f["final_idx"],
                                    v_out, flags)
        out.update(g)
        v_colors = g["v_colors"]
        if s.sh_coeffs is not None:
            v_rgb = v_colors * (sh_rgb + 0.5 > 0).float()  # clamp_min backward
            out["v_coeffs"] = cabi.sh_backward(s.degrees_to_use, s.K, dirs, v_rgb.contiguous())
        pb = cabi.project_backward(cam, means, scales, quats, p["radii"], g["v_xy"], g["v_conic"],
                                   None, vm_dev, pm_dev)
        out.update(pb)
    torch.cuda.synchronize()
    return out


def np_(t):
    return t.detach().cpu().numpy()


def oracle_raster(O, s, xys, conics, colors, cov2d3, depths, v_out=None):
    """Compositing stage of the oracle on given 2-D inputs (cov2d3 = [N,3] xx,xy,yy)."""
    N = len(xys)
    c2 = np.zeros((N, 2, 2), dtype=np.float32)
    c2[:, 0, 0] = cov2d3[:, 0]
    c2[:, 0, 1] = c2[:, 1, 0] = cov2d3[:, 1]
    c2[:, 1, 1] = cov2d3[:, 2]
    f = O.rasterize_forward(s.W, s.H, xys, conics, colors, s.opacities, s.background, c2, depths)
    g = None
    if v_out is not None:
        g = O.rasterize_backward(s.W, s.H, xys, conics, colors, s.opacities, s.background, c2,
                                 depths, f["final_Ts"], f["state"], v_out)
    else:
        O.rasterize_free(f["state"])
    return f, g


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def torch_main_loss(window_1d, rendered, gt, ssim_weight):
    """Model::mainLoss as the torch ops the reference issues (ssim.cpp:7-33, model.cpp:54-56,
    780-784), differentiable: what OpenSplat's GPU build executes on the device."""
    import torch
    import torch.nn.functional as F

    w2 = torch.outer(window_1d, window_1d)[None, None].expand(3, 1, 11, 11).contiguous()
    img1 = gt.permute(2, 0, 1)[None]
    img2 = rendered.permute(2, 0, 1)[None]
    conv = lambda a: F.conv2d(a, w2, padding=5, groups=3)
    mu1, mu2 = conv(img1), conv(img2)
    mu1Sq, mu2Sq, mu1mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1, s2, s12 = conv(img1 * img1) - mu1Sq, conv(img2 * img2) - mu2Sq, conv(img1 * img2) - mu1mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2.0 * mu1mu2 + C1) * (2.0 * s12 + C2)) / ((mu1Sq + mu2Sq + C1) * (s1 + s2 + C2))
    return (1.0 - ssim_weight) * torch.abs(gt - rendered).mean() + ssim_weight * (1.0 - m.mean())
