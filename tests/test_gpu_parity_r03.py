"""-m gpu: the parity configurations round 2 left untested (VERDICT r02 "Next round" item 1).

  (a) SH degree 4 (K = 25; gsplat_cpu.cpp:409-486 returns 25 bases for any degree > 3): the stage
      kernels gs_sh_forward / gs_sh_backward, their fused-glue variants and the fused per-Gaussian
      kernels gs_gaussian_forward/backward<25> against the oracle's values (tests/
      test_gpu_baseline_parity.py adds (25, 4) to the whole-chain test of the timed path);
  (b) GS_FLAG_FAST_EXP (v_exp_f32 instead of the glibc-bit-exact exponential; a shipped, benchmarked
      mode) at full C2 against the oracle chain, with explicit bounds;
  (c) the two behaviours the product takes from the reference's GPU path instead of gsplat-cpu
      (DESIGN.md P2, P3): Gaussians at or behind the near plane (forward.cu:49-52) and a principal
      point off the image centre (helpers.cuh:13-15,112-122), against those formulas restated in the
      oracle (orc_project_gpu_semantics) — stage kernels, the timed fused path and the C++ operators.

Measured values are written to gpurun_out/parity_r03.json (copied to profiles/ by the round script).
"""
import json
import os

import numpy as np
import pytest

from opensplat_amd import scenes
from tests.test_gpu_baseline_parity import (image_flips, oracle_chain, records_of, run_timed_path,
                                            timed_path_grads)
from tests.util import hip_pipeline, np_, rel_err, to_dev

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = {}


def _report(name, **kv):
    REPORT[name] = {k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in kv.items()}
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_r03.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


# ---- (a) SH degree 4 ---------------------------------------------------------------------------------

@pytest.mark.parametrize("deg", [4, 3, 0])
def test_sh_k25_stage_kernels_match_oracle(deg, restated):
    """K = 25 coefficient tensors, degree 4 (all 25 bases) and lower degrees_to_use on the same
    tensor (model.cpp:178 raises the degree during training)."""
    from opensplat_amd import cabi

    s = scenes.camera_scene(10007, 64, 64, K=25, seed=12, yaw_deg=5.0)
    dirs, coeffs = to_dev(s.dirs), to_dev(s.sh_coeffs)
    c = cabi.sh_forward(deg, dirs, coeffs)
    ref = restated.sh_forward(deg, s.dirs, s.sh_coeffs)
    assert np.abs(np_(c) - ref).max() < 2e-6
    v = np.random.RandomState(deg).randn(s.N, 3).astype(np.float32)
    g = cabi.sh_backward(deg, 25, dirs, to_dev(v))
    gref = restated.sh_backward(deg, s.dirs, s.sh_coeffs, v)
    assert np.abs(np_(g) - gref).max() < 1e-6
    nb = (deg + 1) ** 2
    assert np.all(np_(g)[:, nb:, :] == 0)
    if deg == 4:
        assert np.abs(gref[:, 16:, :]).max() > 0.1     # the degree-4 rows carry signal


def test_sh_k25_fused_glue_kernels_match_oracle(restated):
    """gs_sh_forward_fused / gs_sh_backward_fused at K = 25 (cat, view directions, +0.5, clamp_min
    inside the kernel; model.cpp:114,176-177,192) against the oracle with the glue in numpy."""
    from opensplat_amd import cabi

    s = scenes.camera_scene(6001, 64, 64, K=25, seed=13, yaw_deg=-6.0)
    s.sh_coeffs[::3, 0, :] -= 1.2                      # a third of the colours clamp at 0
    cam_pos = np.array([0.3, -0.2, 0.1], np.float32)
    d = (s.means - cam_pos).astype(np.float32)
    dirs = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    means = to_dev(s.means)
    dc, rest = to_dev(s.sh_coeffs[:, 0, :]), to_dev(s.sh_coeffs[:, 1:, :])
    col, rgb = cabi.sh_forward_fused(4, means, cam_pos, dc, rest)
    ref = restated.sh_forward(4, dirs, s.sh_coeffs)
    assert np.abs(np_(rgb) - ref).max() < 3e-6        # the in-kernel direction differs by an ulp
    assert np.array_equal(np_(col), np.maximum(np_(rgb) + np.float32(0.5), 0))
    assert 0.05 < (np_(col) == 0).mean() < 0.6
    v = np.random.RandomState(4).randn(s.N, 3).astype(np.float32)
    v_dc, v_rest = cabi.sh_backward_fused(4, 25, means, cam_pos, rgb, to_dev(v))
    mask = (np_(rgb) + np.float32(0.5) >= 0)           # the kernel's own clamp mask (raw rgb given)
    gref = restated.sh_backward(4, dirs, s.sh_coeffs, (v * mask).astype(np.float32))
    assert np.abs(np_(v_dc) - gref[:, 0, :]).max() < 2e-6
    assert np.abs(np_(v_rest) - gref[:, 1:, :]).max() < 2e-6


def test_operator_chain_degree4_matches_oracle(restated):
    """ProjectGaussians -> SphericalHarmonics(degree 4) -> RasterizeGaussians through the C++ autograd
    operators, all six gradients against the oracle chain."""
    import torch

    from opensplat_amd import ops

    s = scenes.camera_scene(5000, 256, 160, K=25, seed=14, znear=1.0, zfar=100.0, yaw_deg=2.0)
    t = lambda a, rg=False: to_dev(a).requires_grad_(rg)
    means, scales, quats = t(s.means, True), t(s.scales, True), t(s.quats, True)
    coeffs, opac = t(s.sh_coeffs, True), t(s.opacities, True)
    p = ops.project_gaussians(means, scales, 1.0, quats, t(s.viewmat), t(s.projmat), s.fx, s.fy,
                              s.cx, s.cy, s.H, s.W)
    rgb = torch.clamp_min(ops.spherical_harmonics(4, t(s.dirs), coeffs) + 0.5, 0.0)
    img = ops.rasterize_gaussians(p[0], p[1], p[2], p[3], p[4], rgb, opac, s.H, s.W,
                                  t(s.background), p[6])
    img.backward(t(s.v_out))
    torch.cuda.synchronize()
    ref = oracle_chain(restated, s)
    flips, dmax = image_flips(np_(img), ref["img"])
    assert flips <= 4, (flips, dmax)
    errs = dict(v_means=rel_err(np_(means.grad), ref["v_means"]),
                v_scales=rel_err(np_(scales.grad), ref["v_scales"]),
                v_quats=rel_err(np_(quats.grad), ref["v_quats"]),
                v_opacity=rel_err(np_(opac.grad).ravel(), ref["v_opacity"].ravel()),
                v_coeffs=rel_err(np_(coeffs.grad), ref["v_coeffs"]))
    _report("operators_K25_deg4", image_flipped_pixels=flips, image_max_abs_err=dmax,
            **{"rel_" + k: v for k, v in errs.items()})
    for k, e in errs.items():
        assert e < 2e-5, (k, e)
    assert np.abs(np_(coeffs.grad)[:, 16:, :]).max() > 0


# ---- (b) GS_FLAG_FAST_EXP --------------------------------------------------------------------------

# Bounds of the fast-exponential mode (v_exp_f32, <= 1 ulp + the 1-ulp argument scaling, instead of the
# correctly rounded glibc algorithm).  The image can no longer be bit-exact: every alpha is off by
# ~1e-7 relative, which moves a pixel by a few 1e-7, and an alpha within that distance of 1/255 (or a T
# within it of 1e-4) changes the contributor set of that pixel.  Such a "flipped" pixel moves by up to
# alpha * T * |colour| <= 1.5 / 255, and the Gaussian that entered or left it gains or loses one whole
# term of its gradient sums — with the unit-scale cotangent of the benchmark a term is ~1e-3 of the
# largest gradient, so the gradient bound is set by the flips, not by the arithmetic.
# Measured at C2 (profiles/parity_r03.json): 1 flipped pixel of 2 073 600 (1.4e-3), every other pixel
# within 3.6e-7; gradients 2.9e-4 ... 1.1e-3 of max|g| (parity mode: 0 flips, 2.4e-7, 4e-7 ... 8e-7).
# The mode does NOT meet the parity mode's 2e-5 gradient bound; bench.py says so in config.exp.
FAST_EXP_BOUNDS = dict(
    image_max_abs_smooth=2e-6,    # every pixel that kept its contributor set
    flipped_pixels_per_mpix=10,   # pixels above 1e-5 (contributor set changed), per 10^6 pixels
    grad_rel=5e-3,                # each of the six gradient tensors, max|d| / max|ref|
)


def test_fast_exp_mode_at_c2_against_the_oracle_chain(restated):
    from opensplat_amd import cabi

    s = scenes.config_c2()
    pipe = run_timed_path(s, flags=cabi.GS_FLAG_FAST_EXP)
    ref = oracle_chain(restated, s)
    P = s.W * s.H
    img = np_(pipe.fwd["img"])
    d = np.abs(img.astype(np.float64) - ref["img"]).max(axis=-1)
    flips = int((d > 1e-5).sum())
    smooth = float(d[d <= 1e-5].max())
    got = timed_path_grads(pipe)
    errs = {k: rel_err(got[k], ref[k].reshape(got[k].shape))
            for k in ("v_means", "v_scales", "v_quats", "v_opacity", "v_coeffs")}
    # the same comparison for the parity mode, for the record (bit-exact compositing: 0 flips)
    exact = run_timed_path(s, flags=0)
    d0 = np.abs(np_(exact.fwd["img"]).astype(np.float64) - ref["img"]).max(axis=-1)
    _report("fast_exp_c2", pixels=P, image_flipped_pixels=flips, image_max_abs_err=float(d.max()),
            image_max_abs_err_unflipped=smooth, parity_mode_flipped_pixels=int((d0 > 1e-5).sum()),
            parity_mode_image_max_abs_err=float(d0.max()),
            fast_vs_exact_pixels_differing=int((img != np_(exact.fwd["img"])).any(-1).sum()),
            **{"rel_" + k: v for k, v in errs.items()})
    assert smooth < FAST_EXP_BOUNDS["image_max_abs_smooth"], smooth
    assert flips <= FAST_EXP_BOUNDS["flipped_pixels_per_mpix"] * P / 1e6, flips
    assert float(d.max()) < 1.5 / 255 * 1.2, float(d.max())        # a flip moves one contribution
    for k, e in errs.items():
        assert e < FAST_EXP_BOUNDS["grad_rel"], (k, e)


# ---- (c) near-plane cull and principal point -------------------------------------------------------

def _pp_scene(K=16, deg=3, N=12000, W=400, H=240, yaw=0.0, seed=71):
    """Principal point 13.25 px right of / 7.5 px above the image centre; a fifth of the Gaussians at or
    behind the near plane (clip_thresh = 0.01): behind the camera, between 0 and clip, exactly at
    clip (culled: `<=`, helpers.cuh:229), and a few just in front of it (kept; scaled so that their
    pixel footprint stays what it was)."""
    s = scenes.camera_scene(N, W, H, K=K, seed=seed, znear=0.001, zfar=1000.0, yaw_deg=yaw,
                            degrees_to_use=deg, sigma_px=(0.6, 5.0))
    s.cx, s.cy = W / 2.0 + 13.25, H / 2.0 - 7.5
    z_new = np.full(N, np.nan, np.float32)
    z_new[0::10] = -1.0
    z_new[1::10] = 0.005
    if yaw == 0.0:
        z_new[2::20] = np.float32(0.01)          # == clip_thresh: culled
    z_new[7::20] = 0.011                         # just in front: kept
    sel = ~np.isnan(z_new)
    if yaw == 0.0:
        k = (z_new[sel] / s.means[sel, 2]).astype(np.float32)
        s.means[sel] *= k[:, None]
        s.means[sel, 2] = z_new[sel]             # exact values (identity rotation: p_view.z = z)
        s.scales[sel] *= np.abs(k)[:, None]
    else:                                        # rotate the edits into the camera frame
        R = s.viewmat[:3, :3]
        pv = s.means[sel] @ R.T
        k = (z_new[sel] / pv[:, 2]).astype(np.float32)
        s.means[sel] = ((pv * k[:, None]) @ R).astype(np.float32)
        s.scales[sel] *= np.abs(k)[:, None]
    d = s.means.astype(np.float64)
    s.dirs = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    return s


def oracle_chain_gpu_semantics(O, s):
    """gsplat-cpu chain with the near-plane cull and the principal-point offset of the reference's
    GPU path applied to its projection (orc_project_gpu_semantics); culled Gaussians take no part
    and receive zero gradients."""
    with np.errstate(all="ignore"):
        o = O.project_forward(s.means, s.scales, s.quats, s.viewmat, s.projmat, s.fx, s.fy, s.cx,
                              s.cy, s.H, s.W)
    o = O.project_gpu_semantics(o, s.means, s.viewmat, s.projmat, s.cx, s.cy, s.H, s.W, clip=0.01)
    v = o["visible"]
    sh = O.sh_forward(s.degrees_to_use, s.dirs, s.sh_coeffs)
    colors = np.maximum(sh + np.float32(0.5), 0.0).astype(np.float32)
    f = O.rasterize_forward(s.W, s.H, o["xys"][v], o["conics"][v], colors[v], s.opacities[v],
                            s.background, o["cov2d"][v], o["depths"][v], want_contributors=False)
    g = O.rasterize_backward(s.W, s.H, o["xys"][v], o["conics"][v], colors[v], s.opacities[v],
                             s.background, o["cov2d"][v], o["depths"][v], f["final_Ts"], f["state"],
                             s.v_out)
    N = s.N
    full = {k: np.zeros((N,) + g[k].shape[1:], np.float32) for k in g}
    for k in g:
        full[k][v] = g[k]
    v_rgb = (full["v_colors"] * (sh + np.float32(0.5) > 0)).astype(np.float32)
    v_coeffs = O.sh_backward(s.degrees_to_use, s.dirs, s.sh_coeffs, v_rgb)
    pb = O.project_backward(s.means[v], s.scales[v], s.quats[v], s.viewmat, s.projmat, s.fx, s.fy,
                            s.cx, s.cy, s.H, s.W, g["v_xy"], g["v_conic"])
    out = dict(proj=o, visible=v, img=f["img"], final_Ts=f["final_Ts"], v_coeffs=v_coeffs,
               v_opacity=full["v_opacity"], v_xy=full["v_xy"], v_conic=full["v_conic"])
    for k in ("v_means", "v_scales", "v_quats"):
        a = np.zeros((N, pb[k].shape[1]), np.float32)
        a[v] = pb[k]
        out[k] = a
    return out


@pytest.mark.parametrize("yaw", [0.0, 6.0])
def test_stage_kernels_cull_and_principal_point(yaw, restated):
    s = _pp_scene(yaw=yaw)
    ref = oracle_chain_gpu_semantics(restated, s)
    v = ref["visible"]
    assert 0.1 < (~v).mean() < 0.3
    out = hip_pipeline(s)
    radii = np_(out["radii"])
    assert np.array_equal(radii > 0, v)                                  # forward.cu:49-52
    xys = np_(out["xys"])
    assert rel_err(xys[v], ref["proj"]["xys"][v]) < 2e-6
    # ... which is the GPU path's own expression (helpers.cuh:13-15,112-122) up to fp32 round-off and
    # its 1e-6 in the perspective divide
    # its 1 / (w + 1e-6) against gsplat-cpu's 1 / max(w, 1e-6): <= 1e-6 / clip = 1e-4 relative
    want = ref["proj"]["xys_gpu_formula"][v]
    assert np.all(np.abs(xys[v] - want) <= 2e-3 + 1.2e-4 * (np.abs(want) + max(s.W, s.H)))
    far = v & (np_(out["depths"]) > 1.0)
    assert np.abs(xys[far] - ref["proj"]["xys_gpu_formula"][far]).max() < 2e-3
    flips, dmax = image_flips(np_(out["img"]), ref["img"])
    assert flips <= 4, (flips, dmax)
    errs = {}
    for k in ("v_means", "v_scales", "v_quats", "v_coeffs"):
        errs[k] = rel_err(np_(out[k]), ref[k].reshape(np_(out[k]).shape))
    errs["v_opacity"] = rel_err(np_(out["v_opacity"]).ravel(), ref["v_opacity"].ravel())
    _report("stage_kernels_pp_yaw%g" % yaw, culled=int((~v).sum()), image_flipped_pixels=flips,
            image_max_abs_err=dmax, **{"rel_" + k: e for k, e in errs.items()})
    for k, e in errs.items():
        assert e < 2e-5, (k, e)
    for k in ("v_means", "v_scales", "v_quats", "v_opacity"):
        assert not np_(out[k])[~v].any(), k                             # culled: exactly zero


@pytest.mark.parametrize("K,deg", [(16, 3), (1, 0)])
def test_timed_fused_path_cull_and_principal_point(K, deg, restated):
    s = _pp_scene(K=K, deg=deg, yaw=-4.0, seed=72)
    ref = oracle_chain_gpu_semantics(restated, s)
    v = ref["visible"]
    pipe = run_timed_path(s)
    assert np.array_equal(np_(pipe.gfwd["radii"]) > 0, v)
    flips, dmax = image_flips(np_(pipe.fwd["img"]), ref["img"])
    assert flips <= 4, (flips, dmax)
    got = timed_path_grads(pipe)
    errs = {k: rel_err(got[k], ref[k].reshape(got[k].shape))
            for k in ("v_means", "v_scales", "v_quats", "v_opacity", "v_coeffs")}
    _report("fused_pp_K%d" % K, culled=int((~v).sum()), image_flipped_pixels=flips,
            image_max_abs_err=dmax, **{"rel_" + k: e for k, e in errs.items()})
    for k, e in errs.items():
        assert e < 2e-5, (k, e)
    for k in ("v_means", "v_scales", "v_quats", "v_opacity", "v_coeffs"):
        assert not got[k][~v].any(), k
    rec = records_of(pipe, s.N)
    assert rel_err(rec["v_xy"], ref["v_xy"]) < 2e-5


def test_operators_cull_and_principal_point(restated):
    import torch

    from opensplat_amd import ops

    s = _pp_scene(K=4, deg=1, N=6000, W=203, H=117, yaw=3.0, seed=73)
    ref = oracle_chain_gpu_semantics(restated, s)
    v = ref["visible"]
    t = lambda a, rg=False: to_dev(a).requires_grad_(rg)
    means, scales, quats = t(s.means, True), t(s.scales, True), t(s.quats, True)
    coeffs, opac = t(s.sh_coeffs, True), t(s.opacities, True)
    p = ops.project_gaussians(means, scales, 1.0, quats, t(s.viewmat), t(s.projmat), s.fx, s.fy,
                              s.cx, s.cy, s.H, s.W)
    assert np.array_equal(np_(p[2]) > 0, v)
    rgb = torch.clamp_min(ops.spherical_harmonics(s.degrees_to_use, t(s.dirs), coeffs) + 0.5, 0.0)
    img = ops.rasterize_gaussians(p[0], p[1], p[2], p[3], p[4], rgb, opac, s.H, s.W,
                                  t(s.background), p[6])
    img.backward(t(s.v_out))
    torch.cuda.synchronize()
    flips, dmax = image_flips(np_(img), ref["img"])
    assert flips <= 4, (flips, dmax)
    for name, got, want in [("v_means", means.grad, ref["v_means"]), ("v_scales", scales.grad, ref["v_scales"]),
                            ("v_quats", quats.grad, ref["v_quats"]),
                            ("v_opacity", opac.grad.reshape(-1), ref["v_opacity"].reshape(-1)),
                            ("v_coeffs", coeffs.grad, ref["v_coeffs"])]:
        assert rel_err(np_(got), want.reshape(np_(got).shape)) < 2e-5, name
        assert not np_(got)[~v].any(), name
