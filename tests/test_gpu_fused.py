"""-m gpu: SURVEY.md §8 row f1 — the fused-glue kernels and the SplatRender operator against the
unfused chain (the reference's own operator sequence with torch element-wise glue, model.cpp:114-222),
which the other GPU tests tie to the CPU oracle."""
import numpy as np
import pytest

from opensplat_amd import scenes
from tests.util import np_, rel_err, to_dev

pytestmark = pytest.mark.gpu


def _raw_params(s, seed=0):
    """Raw optimiser parameters that reproduce scene `s` through Model::forward's glue."""
    rs = np.random.RandomState(seed)
    log_scales = np.log(s.scales).astype(np.float32)
    quats_raw = (s.quats * rs.uniform(0.5, 2.0, (s.N, 1))).astype(np.float32)
    o = np.clip(s.opacities.reshape(-1, 1), 1e-6, 1 - 1e-6)
    logits = np.log(o / (1 - o)).astype(np.float32)
    dc = np.ascontiguousarray(s.sh_coeffs[:, 0, :])
    rest = np.ascontiguousarray(s.sh_coeffs[:, 1:, :])
    R, t = s.viewmat[:3, :3], s.viewmat[:3, 3]
    cam_pos = (-R.T @ t).astype(np.float32)
    return log_scales, quats_raw, logits, dc, rest, cam_pos


def _unfused(s, P, v_img):
    """OpenSplat's Model::forward (GPU branch) with this repo's three operators."""
    import torch

    from opensplat_amd import ops

    means, ls, q, lo, dc, rest = P
    cam_pos = to_dev(_raw_params(s)[5])
    p = ops.project_gaussians(means, torch.exp(ls), 1.0, q / q.norm(2, -1, True), to_dev(s.viewmat),
                              to_dev(s.projmat), s.fx, s.fy, s.cx, s.cy, s.H, s.W)
    xys = p[0]
    xys.retain_grad()
    colors = torch.cat([dc[:, None, :], rest], 1)
    dirs = means.detach() - cam_pos
    dirs = dirs / dirs.norm(2, -1, True)
    rgbs = ops.spherical_harmonics(s.degrees_to_use, dirs, colors)
    rgbs = torch.clamp_min(rgbs + 0.5, 0.0)
    img = ops.rasterize_gaussians(xys, p[1], p[2], p[3], p[4], rgbs, torch.sigmoid(lo), s.H, s.W,
                                  to_dev(s.background), p[6])
    img = torch.clamp_max(img, 1.0)
    img.backward(v_img)
    return img, xys, p[2]


@pytest.mark.parametrize("K,deg", [(16, 3), (4, 1), (1, 0), (9, 2)])
def test_splat_render_matches_the_unfused_operator_chain(K, deg):
    import torch

    from opensplat_amd import ops

    s = scenes.camera_scene(6000, 320, 200, K=K, seed=51, znear=1.0, zfar=100.0, yaw_deg=4.0,
                            degrees_to_use=deg)
    # brighten so that clamp_max(rgb, 1) is active on part of the image
    s.sh_coeffs[:, 0, :] += 1.0
    raw = _raw_params(s)
    v_img = to_dev(np.random.RandomState(5).uniform(-1, 1, (s.H, s.W, 3)).astype(np.float32))

    def leaves():
        arrs = [s.means, raw[0], raw[1], raw[2], raw[3], raw[4]]
        return [to_dev(a).requires_grad_(True) for a in arrs]

    A = leaves()
    img_a, xys_a, radii_a = _unfused(s, A, v_img)
    B = leaves()
    xys_grad = torch.zeros((s.N, 2), device="cuda")
    out = ops.splat_render(B[0], B[1], B[2], B[3], B[4], B[5] if K > 1 else torch.empty(0, device="cuda"),
                           to_dev(s.viewmat), to_dev(s.projmat), to_dev(raw[5]), s.fx, s.fy, s.cx, s.cy,
                           s.H, s.W, s.degrees_to_use, to_dev(s.background), xys_grad)
    out[0].backward(v_img)
    torch.cuda.synchronize()
    assert (np_(img_a) >= 1.0).mean() > 0.001                     # the clamp is exercised
    assert np.array_equal(np_(out[2]), np_(radii_a))
    assert rel_err(np_(out[1]), np_(xys_a)) < 1e-6
    # the fused sigmoid / exp differ from torch's by an ulp: a handful of threshold flips at most
    d = np.abs(np_(out[0]) - np_(img_a)).max(axis=2)
    assert (d > 1e-5).sum() <= max(4, 2e-5 * d.size), (d > 1e-5).sum()
    names = ["means", "log_scales", "quats", "opacity_logits", "features_dc", "features_rest"]
    for n, a, b in zip(names, A, B):
        if n == "features_rest" and K == 1:
            continue
        ga, gb = np_(a.grad), np_(b.grad)
        assert rel_err(gb, ga) < 2e-3, n
    assert rel_err(np_(xys_grad), np_(xys_a.grad)) < 2e-3


def test_fused_sh_kernels_equal_cat_dirs_sh_clamp():
    import torch

    from opensplat_amd import cabi

    s = scenes.camera_scene(10007, 64, 64, K=16, seed=53, yaw_deg=-7.0)
    raw = _raw_params(s)
    means, dc, rest = to_dev(s.means), to_dev(raw[3]), to_dev(raw[4])
    for deg in [0, 1, 2, 3]:
        col, rgb = cabi.sh_forward_fused(deg, means, raw[5], dc, rest)
        dirs = means - to_dev(raw[5])
        dirs = (dirs / dirs.norm(2, -1, True)).contiguous()
        ref = cabi.sh_forward(deg, dirs, torch.cat([dc[:, None, :], rest], 1).contiguous())
        assert np.abs(np_(rgb) - np_(ref)).max() < 2e-6
        assert np.array_equal(np_(col), np.maximum(np_(rgb) + np.float32(0.5), 0))
        v = to_dev(np.random.RandomState(deg).randn(s.N, 3).astype(np.float32))
        v_dc, v_rest = cabi.sh_backward_fused(deg, 16, means, raw[5], rgb, v)
        gref = cabi.sh_backward(deg, 16, dirs, (v * (rgb + 0.5 >= 0)).contiguous())
        assert np.abs(np_(v_dc) - np_(gref)[:, 0, :]).max() < 1e-6
        assert np.abs(np_(v_rest) - np_(gref)[:, 1:, :]).max() < 1e-6


def test_log_scale_projection_equals_exp_then_project():
    import torch

    from opensplat_amd import cabi

    s = scenes.camera_scene(5000, 203, 117, K=4, seed=57, znear=1.0, zfar=100.0)
    raw = _raw_params(s)
    means, quats = to_dev(s.means), to_dev(raw[1])
    ls = to_dev(raw[0])
    cam0 = cabi.make_camera(s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy, s.W, s.H)
    cam1 = cabi.make_camera(s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy, s.W, s.H,
                            flags=cabi.GS_CAM_LOG_SCALES)
    a = cabi.project_forward(cam0, means, torch.exp(ls).contiguous(), quats)
    b = cabi.project_forward(cam1, means, ls, quats)
    for k in ["xys", "conics", "depths", "cov2d"]:
        assert rel_err(np_(b[k]), np_(a[k])) < 1e-5, k
    rs = np.random.RandomState(1)
    v_xy, v_conic = to_dev(rs.randn(s.N, 2).astype(np.float32)), to_dev(rs.randn(s.N, 3).astype(np.float32))
    ga = cabi.project_backward(cam0, means, torch.exp(ls).contiguous(), quats, a["radii"], v_xy, v_conic)
    gb = cabi.project_backward(cam1, means, ls, quats, b["radii"], v_xy, v_conic)
    assert rel_err(np_(gb["v_means"]), np_(ga["v_means"])) < 1e-4
    assert rel_err(np_(gb["v_quats"]), np_(ga["v_quats"])) < 1e-4
    assert rel_err(np_(gb["v_scales"]), np_(ga["v_scales"] * torch.exp(ls))) < 1e-4


# ---- fused per-Gaussian stages (gs_gaussian_forward / gs_gaussian_backward) ----------------------

@pytest.mark.parametrize("K,deg,N", [(16, 3, 5000), (4, 1, 3001), (1, 0, 777), (9, 2, 64), (25, 4, 1500)])
def test_gaussian_forward_backward_equal_the_stage_kernels(K, deg, N):
    """One kernel per direction vs the three stage kernels it replaces, same flags: the device
    functions are shared (gs_gaussian.h), so the packed records, depths, radii, raw colours and all
    six gradient tensors must be the same bits."""
    import torch

    from opensplat_amd import cabi

    s = scenes.camera_scene(N, 320, 200, K=K, seed=91, znear=1.0, zfar=100.0, yaw_deg=2.0,
                            degrees_to_use=deg)
    s.sh_coeffs[:, 0, :] += 0.7
    s.means[::7, 2] = -1.0                      # behind the camera: culled
    raw = _raw_params(s)
    means, ls, q, lo = to_dev(s.means), to_dev(raw[0]), to_dev(raw[1]), to_dev(raw[2].reshape(-1))
    dc, rest = to_dev(raw[3]), (to_dev(raw[4]) if K > 1 else None)
    cam = cabi.make_camera(s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy, s.W, s.H,
                           flags=cabi.GS_CAM_LOG_SCALES)
    cam_pos = raw[5]
    flags = cabi.GS_FLAG_LOGIT_OPACITY | cabi.GS_FLAG_CLAMP_IMAGE
    # stage kernels
    p = cabi.project_forward(cam, means, ls, q)
    colors, rgb_raw = cabi.sh_forward_fused(deg, means, cam_pos, dc, rest)
    b = cabi.bin_and_sort(s.W, s.H, p["xys"], p["depths"], p["radii"], p["conics"], colors, lo,
                          p["cov2d"], flags=flags)
    # fused
    g = cabi.gaussian_forward(cam, means, ls, q, lo, dc, rest, cam_pos, deg, flags, want_xys=True)
    torch.cuda.synchronize()
    assert torch.equal(g["packed"], b.packed)
    assert torch.equal(g["rgb_raw"], rgb_raw)
    assert torch.equal(g["depths"], p["depths"]) and torch.equal(g["radii"], p["radii"])
    assert torch.equal(g["xys"], p["xys"])
    assert (g["radii"] == 0).sum() > 0
    b2 = cabi.bin_and_sort(s.W, s.H, None, g["depths"], None, None, None, None, packed=g["packed"])
    assert torch.equal(b2.gaussian_ids_sorted, b.gaussian_ids_sorted) and torch.equal(b2.tile_bins, b.tile_bins)

    f = cabi.rasterize_forward(s.W, s.H, b, s.background, flags,
                               out=dict(img=torch.empty((s.H, s.W, 3), device="cuda"),
                                        final_Ts=torch.empty((s.H, s.W), device="cuda"),
                                        final_idx=torch.empty((s.H, s.W), device="cuda", dtype=torch.int32),
                                        img_clamped=torch.empty((s.H, s.W, 3), device="cuda")))
    v_out = to_dev(np.random.RandomState(4).uniform(-1, 1, (s.H, s.W, 3)).astype(np.float32))
    ws = torch.zeros(cabi.lib().gs_rasterize_backward_workspace_bytes(N) + 64, device="cuda", dtype=torch.uint8)
    # stage path
    gr = cabi.rasterize_backward(s.W, s.H, N, b, s.background, f["final_Ts"], f["final_idx"], v_out,
                                 flags, workspace=ws, img_raw=f["img"])
    v_dc, v_rest = cabi.sh_backward_fused(deg, K, means, cam_pos, rgb_raw, gr["v_colors"])
    pb = cabi.project_backward(cam, means, ls, q, p["radii"], gr["v_xy"], gr["v_conic"])
    # fused path: records kept, consumed and zeroed by gaussian_backward
    ws2 = torch.zeros_like(ws)
    cabi.rasterize_backward(s.W, s.H, N, b, s.background, f["final_Ts"], f["final_idx"], v_out,
                            flags | cabi.GS_FLAG_KEEP_RECORDS | cabi.GS_FLAG_RECORDS_ZEROED,
                            workspace=ws2, img_raw=f["img"])
    fz = dict(device="cuda", dtype=torch.float32)
    out = dict(v_means=torch.empty((N, 3), **fz), v_scales=torch.empty((N, 3), **fz),
               v_quats=torch.empty((N, 4), **fz), v_opacity=torch.empty((N,), **fz),
               v_dc=torch.empty((N, 3), **fz), v_rest=torch.empty((N, max(K - 1, 1), 3), **fz))
    v_xy = torch.empty((N, 2), **fz)
    cabi.gaussian_backward(cam, means, ls, q, lo, cam_pos, K, deg, g["radii"], g["rgb_raw"], ws2, out,
                           flags | cabi.GS_FLAG_RECORDS_ZEROED, v_xy=v_xy)
    torch.cuda.synchronize()
    # the compositing backward accumulates with atomics: two runs differ in the last bits, so the
    # comparison is at summation-order tolerance, and exact where no atomics are involved
    def close(a, b_, name):
        a, b_ = np_(a), np_(b_)
        assert rel_err(a, b_) < 2e-5, name
    close(v_xy, gr["v_xy"], "v_xy")
    close(out["v_opacity"], gr["v_opacity"], "v_opacity")
    close(out["v_dc"], v_dc, "v_dc")
    if K > 1:
        close(out["v_rest"][:, :K - 1], v_rest, "v_rest")
    close(out["v_means"], pb["v_means"], "v_means")
    close(out["v_scales"], pb["v_scales"], "v_scales")
    close(out["v_quats"], pb["v_quats"], "v_quats")
    assert not ws2[: N * 64].any()            # GS_FLAG_RECORDS_ZEROED: records zeroed behind the read
    # determinism of the non-atomic part: feed the SAME records to both consumers
    ws3 = ws.clone()                           # records of the stage run (not zeroed by it)
    out2 = {k: torch.empty_like(v) for k, v in out.items()}
    cabi.gaussian_backward(cam, means, ls, q, lo, cam_pos, K, deg, g["radii"], g["rgb_raw"], ws3, out2, flags)
    torch.cuda.synchronize()
    assert torch.equal(ws3, ws)                # without the flag the records are left alone
    assert torch.equal(out2["v_means"], pb["v_means"]) and torch.equal(out2["v_scales"], pb["v_scales"])
    assert torch.equal(out2["v_quats"], pb["v_quats"]) and torch.equal(out2["v_dc"], v_dc)
    assert torch.equal(out2["v_opacity"], gr["v_opacity"])
    if K > 1:
        assert torch.equal(out2["v_rest"][:, :K - 1], v_rest)


@pytest.mark.parametrize("K,deg", [(16, 3), (16, 2), (4, 1), (1, 0), (9, 2), (25, 4)])
def test_sh_backward_cameras_equals_the_sum_of_per_camera_sh_gradients(K, deg):
    """gs_sh_backward_cameras (the factored gradient exchange): SH gradients formed from the colour
    cotangents of several cameras == the per-camera SH gradients of gs_sh_backward_fused, summed;
    rows whose cotangent is zero for a camera contribute nothing."""
    import torch

    from opensplat_amd import cabi

    rs = np.random.RandomState(11)
    N, C = 3001, 3
    means = rs.uniform(-2, 2, (N, 3)).astype(np.float32)
    cams = rs.uniform(-6, 6, (C, 3)).astype(np.float32)
    vcol = rs.normal(size=(C, N, 3)).astype(np.float32)
    vcol[1, ::3] = 0.0           # unseen from camera 1
    rgb_raw = np.ones((N, 3), np.float32)   # nothing clamped: the mask was applied by the sender
    dm, dv = to_dev(means), to_dev(vcol)
    want_dc = np.zeros((N, 3), np.float64)
    want_rest = np.zeros((N, max(K - 1, 0), 3), np.float64)
    for c in range(C):
        v_dc = torch.empty((N, 3), device="cuda")
        v_rest = torch.empty((N, max(K - 1, 1), 3), device="cuda")
        cabi.sh_backward_fused(deg, K, dm, to_dev(cams[c]), to_dev(rgb_raw), dv[c].contiguous(),
                               out=(v_dc, v_rest))
        want_dc += np_(v_dc)
        if K > 1:
            want_rest += np_(v_rest)
    # camera centres 4 floats apart, cotangents with a padded stride (as in the all-gather message)
    cp = torch.zeros((C, 4), device="cuda")
    cp[:, :3] = to_dev(cams)
    stride = N * 3 + 8
    flat = torch.zeros(C * stride, device="cuda")
    for c in range(C):
        flat[c * stride:c * stride + N * 3] = dv[c].reshape(-1)
    got_dc = torch.full((N, 3), 7.0, device="cuda")
    got_rest = torch.full((N, max(K - 1, 1), 3), 7.0, device="cuda")
    cabi.sh_backward_cameras(K, deg, dm, cp, flat, got_dc, got_rest, v_colors_stride=stride)
    assert np.abs(np_(got_dc) - want_dc).max() <= 2e-6 * np.abs(want_dc).max()
    if K > 1:
        assert np.abs(np_(got_rest) - want_rest).max() <= 2e-6 * max(np.abs(want_rest).max(), 1e-30)
    # accumulate on top
    cabi.sh_backward_cameras(K, deg, dm, cp, flat, got_dc, got_rest, cabi.GS_FLAG_ACCUMULATE_GRADS,
                             v_colors_stride=stride)
    assert np.abs(np_(got_dc) - 2 * want_dc).max() <= 4e-6 * np.abs(want_dc).max()


def test_sh_backward_cameras_against_the_oracle(restated):
    """The same against the CPU oracle's SH backward (gsplat_cpu.cpp:436-483 restated), camera by
    camera and summed in float64 — not only against this repo's own single-camera kernel."""
    import torch

    from opensplat_amd import cabi

    rs = np.random.RandomState(12)
    N, C, K, deg = 1500, 4, 16, 3
    means = rs.uniform(-2, 2, (N, 3)).astype(np.float32)
    cams = rs.uniform(-6, 6, (C, 3)).astype(np.float32)
    vcol = rs.normal(size=(C, N, 3)).astype(np.float32)
    coeffs = np.zeros((N, K, 3), np.float32)
    want = np.zeros((N, K, 3), np.float64)
    for c in range(C):
        d = means - cams[c]
        dirs = (d / np.sqrt((d.astype(np.float32) ** 2).sum(1, dtype=np.float32))[:, None]).astype(np.float32)
        want += restated.sh_backward(deg, dirs, coeffs, vcol[c])
    cp = torch.zeros((C, 4), device="cuda")
    cp[:, :3] = to_dev(cams)
    v_dc = torch.empty((N, 3), device="cuda")
    v_rest = torch.empty((N, K - 1, 3), device="cuda")
    cabi.sh_backward_cameras(K, deg, to_dev(means), cp, to_dev(vcol), v_dc, v_rest)
    got = np.concatenate([np_(v_dc)[:, None, :], np_(v_rest)], axis=1)
    # view directions are normalised in fp32 on both sides; the basis amplifies an ulp of the direction
    assert np.abs(got - want).max() <= 5e-6 * np.abs(want).max()
