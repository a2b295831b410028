#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from OpenSplat's OWN CPU implementation.

Runs only in the build container, where /root/reference exists and `make -C oracle ref` has
compiled rasterizer/gsplat-cpu + the three op wrappers in place into oracle/_ref/libgsplat_ref.so.
The fixtures hold seeded inputs' *outputs* (the inputs are regenerated from the seed by
opensplat_amd/scenes.py, and a checksum of them is stored to catch generator drift):

  ref_c1_small.npz      simple_trainer set-up (simple_trainer.cpp:79-146), N=600, 64x48, no SH
  ref_camera_sh.npz     perspective camera, SH degree 3 (K=16), N=800, 80x56
  ref_c1_known.npz      the full BASELINE config 1 (N=10 000, 256x256): iteration-1 loss, image
                        mean/sum and max|grad| of simple_trainer.cpp's MSE set-up
  ref_c1_grads.npz      the same run's COMPLETE gradient tensors (means, scales, quats, colours,
                        opacities) and image: what north_star's "gradient max-abs-error < 1e-4 vs the
                        CPU reference" is asserted against on the GPU (python make_golden.py c1grads)

`img`, `final_Ts`, `contributors`, `rast_*` are the reference's *_tensor_cpu functions fed TRUE
depths (contiguous camDepths); `chain_*` is the reference's end-to-end op chain, which sorts by
`proj_depth_keys_as_read` instead.

Usage: python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from opensplat_amd import scenes  # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def scene_digest(s):
    parts = [s.means, s.scales, s.quats, s.opacities, s.viewmat, s.projmat]
    if s.sh_coeffs is not None:
        parts += [s.sh_coeffs, s.dirs]
    if s.colors is not None:
        parts += [s.colors]
    return digest(*parts)


def stages(R, s, v_out):
    p = R.project_forward(s.means, s.scales, s.quats, s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy,
                          s.H, s.W)
    out = {"proj_" + k: v for k, v in p.items()}
    if s.sh_coeffs is not None:
        sh = R.sh_forward(s.degrees_to_use, s.dirs, s.sh_coeffs)
        out["sh_rgb"] = sh
        colors = np.maximum(sh + 0.5, 0.0).astype(np.float32)
    else:
        colors = s.colors
    f = R.rasterize_forward(s.W, s.H, p["xys"], p["conics"], colors, s.opacities, s.background,
                            p["cov2d"], p["cam_depths"])
    out.update(img=f["img"], final_Ts=f["final_Ts"], px_counts=f["px_counts"],
               contributors=f["contributors"])
    g = R.rasterize_backward(s.W, s.H, p["xys"], p["conics"], colors, s.opacities, s.background,
                             p["cov2d"], p["cam_depths"], f["final_Ts"], f["state"], v_out)
    out.update({"rast_" + k: v for k, v in g.items()})
    pb = R.project_backward(s.means, s.scales, s.quats, s.viewmat, s.projmat, s.fx, s.fy, s.cx,
                            s.cy, s.H, s.W, g["v_xy"], g["v_conic"])
    out.update({"proj_" + k: v for k, v in pb.items()})
    if s.sh_coeffs is not None:
        vrgb = (g["v_colors"] * (out["sh_rgb"] + 0.5 > 0)).astype(np.float32)
        out["sh_v_coeffs"] = R.sh_backward(s.degrees_to_use, s.dirs, s.sh_coeffs, vrgb)
    # whole chain through the reference's op wrappers + libtorch autograd.  NB the reference's CPU
    # chain composites in the order of `proj_depth_keys_as_read` (its strided-view quirk, DESIGN.md
    # P11), NOT in depth order, so chain_img != img in general.
    ch = R.chain_fwd_bwd(s.means, s.scales, s.quats, s.dirs, s.sh_coeffs if s.sh_coeffs is not None
                         else s.colors, s.opacities, s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy,
                         s.H, s.W, s.background, v_out, degrees_to_use=s.degrees_to_use)
    for k in ["img", "v_means", "v_scales", "v_quats", "v_coeffs", "v_opacities"]:
        out["chain_" + k] = ch[k]
    out["scene_sha256"] = np.frombuffer(scene_digest(s).encode(), dtype=np.uint8)
    out["v_out"] = v_out
    return out


def c1_grads(R):
    """BASELINE config 1, iteration 1 of simple_trainer.cpp through the reference's own op chain
    (ProjectGaussiansCPU -> RasterizeGaussiansCPU under libtorch autograd, mean-MSE loss): every
    gradient tensor, complete.  The chain composites in the order of the as-read keys (P11)."""
    s = scenes.config_c1()
    gt = s.extra["gt_image"]
    ch0 = R.chain_fwd_bwd(s.means, s.scales, s.quats, None, s.colors, s.opacities, s.viewmat,
                          s.projmat, s.fx, s.fy, s.cx, s.cy, s.H, s.W, s.background, None)
    img = ch0["img"]
    v_out = (2.0 * (img - gt) / img.size).astype(np.float32)   # d mean((img-gt)^2) / d img
    ch = R.chain_fwd_bwd(s.means, s.scales, s.quats, None, s.colors, s.opacities, s.viewmat,
                         s.projmat, s.fx, s.fy, s.cx, s.cy, s.H, s.W, s.background, v_out)
    np.savez_compressed(
        os.path.join(HERE, "ref_c1_grads.npz"), img=img.astype(np.float16),  # image: statistics only
        v_means=ch["v_means"], v_scales=ch["v_scales"], v_quats=ch["v_quats"],
        v_colors=ch["v_coeffs"], v_opacities=ch["v_opacities"],
        scene_sha256=np.frombuffer(scene_digest(s).encode(), dtype=np.uint8))


def main():
    R = oracle.reference()
    if len(sys.argv) > 1 and sys.argv[1] == "c1grads":
        c1_grads(R)
        return
    s = scenes.simple_trainer_scene(600, 64, 48, seed=0)
    v = np.random.RandomState(100).uniform(-1, 1, (s.H, s.W, 3)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "ref_c1_small.npz"), **stages(R, s, v))

    s = scenes.camera_scene(800, 80, 56, K=16, seed=21, sigma_px=(0.7, 5.0), znear=1.0, zfar=100.0)
    np.savez_compressed(os.path.join(HERE, "ref_camera_sh.npz"), **stages(R, s, s.v_out))

    # BASELINE config 1, iteration 1 of simple_trainer.cpp: loss = MSE(img, gt), grads of the loss
    s = scenes.config_c1()
    gt = s.extra["gt_image"]
    ch0 = R.chain_fwd_bwd(s.means, s.scales, s.quats, None, s.colors, s.opacities, s.viewmat,
                          s.projmat, s.fx, s.fy, s.cx, s.cy, s.H, s.W, s.background, None)
    img = ch0["img"]
    loss = float(np.mean((img.astype(np.float64) - gt) ** 2))
    v_out = (2.0 * (img - gt) / img.size).astype(np.float32)
    ch = R.chain_fwd_bwd(s.means, s.scales, s.quats, None, s.colors, s.opacities, s.viewmat,
                         s.projmat, s.fx, s.fy, s.cx, s.cy, s.H, s.W, s.background, v_out)
    np.savez_compressed(
        os.path.join(HERE, "ref_c1_known.npz"), loss=np.float64(loss),
        img_mean=np.float64(img.astype(np.float64).mean()), img_sum=np.float64(img.astype(np.float64).sum()),
        img_small=img[::8, ::8].copy(),
        max_abs_v_means=np.abs(ch["v_means"]).max(), max_abs_v_scales=np.abs(ch["v_scales"]).max(),
        max_abs_v_quats=np.abs(ch["v_quats"]).max(), max_abs_v_colors=np.abs(ch["v_coeffs"]).max(),
        max_abs_v_opacities=np.abs(ch["v_opacities"]).max(),
        v_means_head=ch["v_means"][:64].copy(), v_scales_head=ch["v_scales"][:64].copy(),
        scene_sha256=np.frombuffer(scene_digest(s).encode(), dtype=np.uint8))
    c1_grads(R)
    print("C1 iteration-1: loss %.9g image mean %.9g sum %.9g" % (loss, img.mean(dtype=np.float64),
                                                                  img.sum(dtype=np.float64)))
    print("max|grad| means %.6g scales %.6g quats %.6g colors(sigmoid'ed) %.6g opac(sigmoid'ed) %.6g" % (
        np.abs(ch["v_means"]).max(), np.abs(ch["v_scales"]).max(), np.abs(ch["v_quats"]).max(),
        np.abs(ch["v_coeffs"]).max(), np.abs(ch["v_opacities"]).max()))


if __name__ == "__main__":
    main()
