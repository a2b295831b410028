"""-m gpu: frames of few tiles — the forward's two-entries-per-step walk (same bits as one entry per
step) and the segmented backward (pieces of a tile's list run side by side from the forward's checkpoint
records; gsplat_hip.h: gs_rasterize_checkpoint_plan / _forward_ckpt / _backward_ckpt) against the one-pass
backward and the CPU oracle (gsplat_cpu.cpp:188-240, :313-373)."""
import ctypes as C

import numpy as np
import pytest

from opensplat_amd import scenes
from tests.util import hip_pipeline, np_, oracle_raster, rel_err, to_dev

pytestmark = pytest.mark.gpu

ILP1, ILP2 = 1 << 23, 2 << 23   # flag bits 23..24: entries per step of the compositing forward


def _deep_scene(seed=19, W=64, H=48, N=6000, opacity=None, K=0):
    s = scenes.camera_scene(N, W, H, K=K, seed=seed, sigma_px=(3.0, 10.0), znear=1.0, zfar=100.0)
    if opacity is not None:
        s.opacities[:] = opacity
    return s


def _front(s, flags=0):
    """project + bin once: what both the plain and the checkpointed compositing runs start from."""
    import torch

    from opensplat_amd import cabi

    cam = cabi.make_camera(s.viewmat, s.projmat, s.fx, s.fy, s.cx, s.cy, s.W, s.H)
    p = cabi.project_forward(cam, to_dev(s.means), to_dev(s.scales), to_dev(s.quats), None, None)
    colors = to_dev(s.colors)
    b = cabi.bin_and_sort(s.W, s.H, p["xys"], p["depths"], p["radii"], p["conics"], colors,
                          to_dev(s.opacities.reshape(-1)), p["cov2d"])
    torch.cuda.synchronize()
    return p, b


def _checkpoints(s, seg_len, max_segments):
    from opensplat_amd import cabi

    ck = cabi.Checkpoints()
    assert ck.plan(s.W, s.H, None, "cuda:0", seg_len=seg_len, max_segments=max_segments)
    ck.buf.fill_(0xFF)    # (NaN everywhere: a record that is read without having been written shows)
    return ck


@pytest.mark.parametrize("make", [
    lambda: scenes.simple_trainer_scene(2500, 96, 96, seed=1),
    lambda: scenes.camera_scene(6000, 240, 135, K=0, seed=12, znear=1.0, zfar=100.0),
    lambda: _deep_scene(opacity=0.95),
    lambda: _deep_scene(seed=5, W=50, H=37, N=3000),
])
def test_two_entries_per_step_composite_the_same_bits(make):
    import torch

    from opensplat_amd import cabi

    s = make()
    p, b = _front(s)
    one = cabi.rasterize_forward(s.W, s.H, b, s.background, ILP1)
    two = cabi.rasterize_forward(s.W, s.H, b, s.background, ILP2)
    auto = cabi.rasterize_forward(s.W, s.H, b, s.background, 0)
    torch.cuda.synchronize()
    for k in ("img", "final_Ts", "final_idx"):
        assert torch.equal(one[k], two[k]), k
        assert torch.equal(one[k], auto[k]), k
    assert int((one["final_idx"] >= 0).sum()) > 0


def test_checkpoint_records_are_the_forward_state_in_front_of_their_entry():
    """Record k of a tile = the compositing of the first k * seg_len entries of its list (as long as no entry
    reaches alpha > 0.99, see k_rasterize_forward): a plain forward over
    the lists cut there must end with the same bits (for every pixel that was still being composited at
    that entry — the others keep their final state and are never asked for the record); record 0 = the end."""
    import torch

    from opensplat_amd import cabi

    s = _deep_scene(opacity=0.35)
    s.background[:] = 0.0   # (then the image IS the colour sum of record 0)
    S, MAXSEG = 64, 12
    p, b = _front(s)
    ck = _checkpoints(s, S, MAXSEG)
    f = cabi.rasterize_forward(s.W, s.H, b, s.background, 0, checkpoints=ck)
    torch.cuda.synchronize()
    tiles_x, tiles_y = (s.W + 15) // 16, (s.H + 15) // 16
    rec = np_(ck.buf[: tiles_x * tiles_y * MAXSEG * 4096]).view(np.float32).reshape(tiles_y, tiles_x, MAXSEG, 16, 16, 4)
    # [tile, k, pixel] -> image layout [k, H, W, 4]
    rec = rec.transpose(2, 0, 3, 1, 4, 5).reshape(MAXSEG, tiles_y * 16, tiles_x * 16, 4)[:, : s.H, : s.W]
    img, Ts, idx = np_(f["img"]), np_(f["final_Ts"]), np_(f["final_idx"])
    # (no entry of this scene reaches alpha > 0.99 — opacity 0.35 — so the records are the forward's own state:
    # the factor between the forward's and the backward's transmittance is 1, the colour sums are the image's)
    assert np.array_equal(rec[0, ..., 0], np.ones_like(Ts))
    assert np.array_equal(rec[0, ..., 1:], img)
    bins = np_(b.tile_bins).copy()
    assert (bins[:, 1] - bins[:, 0]).max() > 3 * S
    start = np.repeat(np.repeat(bins[:, 0].reshape(tiles_y, tiles_x), 16, 0), 16, 1)[: s.H, : s.W]
    checked = 0
    for k in range(1, 6):
        cut = bins.copy()
        cut[:, 1] = np.minimum(cut[:, 1], cut[:, 0] + k * S)
        bc = cabi.Binned(b.packed, b.tiles_hit, b.num_isects, b.gaussian_ids_sorted, to_dev(cut), b.block_masks)
        g = cabi.rasterize_forward(s.W, s.H, bc, s.background, 0)
        torch.cuda.synchronize()
        live = idx >= start + k * S     # the pixel composited something at or behind the record's entry
        checked += int(live.sum())
        assert np.array_equal(rec[k, ..., 0][live], np_(g["final_Ts"])[live]), k
        assert np.array_equal(rec[k, ..., 1:][live], np_(g["img"])[live]), k
    assert checked > 1000


@pytest.mark.parametrize("name,make,seg_len,max_segments", [
    ("deep", lambda: _deep_scene(opacity=0.35), 64, 24),
    ("deep-128", lambda: _deep_scene(opacity=0.35), 128, 12),
    ("saturating", lambda: _deep_scene(opacity=0.95), 64, 24),
    ("outgrown-plan", lambda: _deep_scene(opacity=0.35), 64, 3),       # the last piece takes the rest
    ("ragged", lambda: _deep_scene(seed=5, W=50, H=37, N=3000), 64, 16),
    # random opacities up to 1: entries with alpha > 0.99, behind which the reference's backward (alpha clamped
    # at 0.99, gsplat_cpu.cpp:338) and its forward (0.999, :220) disagree about T — the records carry the former
    ("hot-entries", lambda: _deep_scene(), 64, 64),
    ("hot-entries-128", lambda: _deep_scene(seed=23), 128, 32),
    ("short-lists", lambda: scenes.simple_trainer_scene(2500, 96, 96, seed=1), 64, 4),
])
def test_segmented_backward_matches_one_pass_and_oracle(name, make, seg_len, max_segments, restated):
    import torch

    from opensplat_amd import cabi

    s = make()
    if s.v_out is None:
        s.v_out = np.random.RandomState(7).standard_normal((s.H, s.W, 3)).astype(np.float32)
    p, b = _front(s)
    v_out = to_dev(s.v_out)
    f1 = cabi.rasterize_forward(s.W, s.H, b, s.background, 0)
    g1 = cabi.rasterize_backward(s.W, s.H, s.N, b, s.background, f1["final_Ts"], f1["final_idx"], v_out, 0)
    ck = _checkpoints(s, seg_len, max_segments)
    f2 = cabi.rasterize_forward(s.W, s.H, b, s.background, 0, checkpoints=ck)
    g2 = cabi.rasterize_backward(s.W, s.H, s.N, b, s.background, f2["final_Ts"], f2["final_idx"], v_out, 0,
                                 checkpoints=ck)
    torch.cuda.synchronize()
    for k in ("img", "final_Ts", "final_idx"):
        assert torch.equal(f1[k], f2[k]), k
    fo, go = oracle_raster(restated, s, np_(p["xys"]), np_(p["conics"]), s.colors, np_(p["cov2d"]),
                           np_(p["depths"]), s.v_out)
    assert np.array_equal(np_(f2["img"]), fo["img"])
    for k in ("v_xy", "v_conic", "v_colors", "v_opacity"):
        assert np.isfinite(np_(g2[k])).all(), k
        assert rel_err(np_(g2[k]), np_(g1[k])) < 2e-5, (name, k, "against the one-pass backward")
        assert rel_err(np_(g2[k]), go[k]) < 5e-5, (name, k, "against the oracle")


def test_segmented_backward_is_reproducible_under_the_deterministic_flag():
    import torch

    from opensplat_amd import cabi

    s = _deep_scene(opacity=0.35)
    p, b = _front(s)
    v_out = to_dev(s.v_out)
    ck = _checkpoints(s, 64, 24)
    f = cabi.rasterize_forward(s.W, s.H, b, s.background, 0, checkpoints=ck)
    runs = []
    for _ in range(2):
        g = cabi.rasterize_backward(s.W, s.H, s.N, b, s.background, f["final_Ts"], f["final_idx"], v_out,
                                    cabi.GS_FLAG_DETERMINISTIC, checkpoints=ck)
        torch.cuda.synchronize()
        runs.append({k: v.clone() for k, v in g.items()})
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k


def test_checkpoint_plan_and_argument_checks():
    from opensplat_amd import cabi

    ck = cabi.Checkpoints()
    stats = (C.c_int32 * 2)(60000, 900)
    assert ck.plan(384, 288, stats, "cuda:0")
    assert ck.seg_len >= 64 and ck.seg_len & (ck.seg_len - 1) == 0
    assert ck.max_segments * ck.seg_len >= 900
    assert ck.bytes == 24 * 18 * ck.max_segments * 4096
    assert not ck.plan(1920, 1080, stats, "cuda:0")                         # a full frame: many tiles
    assert ck.plan(1008, 756, (C.c_int32 * 2)(600000, 2000), "cuda:0")      # mid-size, lists far beyond the mean
    assert not ck.plan(1008, 756, (C.c_int32 * 2)(1500000, 1200), "cuda:0") # mid-size, no tail
    assert ck.plan(384, 288, (C.c_int32 * 2)(400000, 1400), "cuda:0")       # far from filling the chip: always
    assert not ck.plan(752, 500, (C.c_int32 * 2)(1500000, 1400), "cuda:0")  # 1504 tiles, even lists
    assert not ck.plan(384, 288, None, "cuda:0")                            # no statistics yet
    assert not ck.plan(384, 288, (C.c_int32 * 2)(5000, 100), "cuda:0")      # nothing worth cutting
    # a buffer that is too small, or a segment length that is not a power of two, is refused
    s = scenes.simple_trainer_scene(500, 64, 64, seed=2)
    p, b = _front(s)
    bad = _checkpoints(s, 64, 4)
    bad.max_segments = 4000
    with pytest.raises(RuntimeError):
        cabi.rasterize_forward(s.W, s.H, b, s.background, 0, checkpoints=bad)
    bad = _checkpoints(s, 64, 4)
    bad.seg_len = 96
    with pytest.raises(RuntimeError):
        cabi.rasterize_forward(s.W, s.H, b, s.background, 0, checkpoints=bad)


def test_trainer_trains_the_same_with_and_without_segments():
    """Twenty iterations on a low-resolution frame with long lists: the checkpoint plan engages (from the second
    iteration on: it needs the first one's list statistics) and the parameters stay within summation-order
    distance of the one-pass run."""
    import math

    import torch

    from opensplat_amd import train
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    from train_synthetic_inputs import ground_truth, make_camera, sfm_like_init

    K, W, H = 4, 96, 72
    dev = torch.device("cuda", 0)
    bg = np.zeros(3, np.float32)
    out = []
    for segmented in (False, True):
        rs = np.random.RandomState(3)
        gt = ground_truth(8000, K, rs)
        cams = [make_camera((3.5 * math.cos(t), 0.3, 3.5 * math.sin(t)), W, H) for t in (0.0, 1.0, 2.0, 3.0)]
        G = train.Trainer(*gt, dev)
        images = [G.render(c, bg, 1).clone() for c in cams]
        T = train.Trainer(*sfm_like_init(gt, 3000, K, rs), dev, max_steps=1000, deterministic=True,
                          segmented=segmented)
        losses = []
        for it in range(1, 21):
            losses.append(T.train_step(cams[it % 4], images[it % 4], bg, 1).clone())
        torch.cuda.synchronize()
        if segmented:
            assert T._ckpt.bytes > 0 and T._ckpt.max_segments >= 3
        out.append(([t.clone() for t in (T.means, T.log_scales, T.quats, T.opacity_logits, T.features_dc)],
                    torch.stack(losses).cpu().numpy()))
    (pa, la), (pb, lb) = out
    assert np.abs(la[:, 0] - lb[:, 0]).max() < 1e-4 * np.abs(la[:, 0]).max(), (la[:, 0], lb[:, 0])
    for a, b_ in zip(pa, pb):
        # (Adam's first steps move a parameter by ~lr whatever its gradient's size, so an element whose gradient
        # is all rounding may go the other way: the bulk must agree, not every element)
        d = (a - b_).abs() / max(float(a.abs().max()), 1.0)
        assert float(d.mean()) < 1e-4 and float((d > 1e-2).float().mean()) < 1e-3


def test_splat_render_uses_the_pieces_from_its_second_frame_on():
    """The C++ operator (what the --fused Model calls): the first frame of a size has no list statistics, from
    the second on the plan engages; the six parameter gradients and d loss / d xys stay within summation
    order of the one-pass backward, the image is the same bits."""
    import torch

    from opensplat_amd import ops
    from tests.test_gpu_fused import _raw_params

    s = scenes.camera_scene(6000, 64, 48, K=4, seed=19, sigma_px=(3.0, 10.0), znear=1.0, zfar=100.0,
                            degrees_to_use=1)
    raw = _raw_params(s)
    v_img = to_dev(np.random.RandomState(5).uniform(-1, 1, (s.H, s.W, 3)).astype(np.float32))

    def run():
        P = [to_dev(a).requires_grad_(True) for a in [s.means, raw[0], raw[1], raw[2], raw[3], raw[4]]]
        xys_grad = torch.zeros((s.N, 2), device="cuda")
        out = ops.splat_render(P[0], P[1], P[2], P[3], P[4], P[5], to_dev(s.viewmat), to_dev(s.projmat),
                               to_dev(raw[5]), s.fx, s.fy, s.cx, s.cy, s.H, s.W, s.degrees_to_use,
                               to_dev(s.background), xys_grad)
        out[0].backward(v_img)
        torch.cuda.synchronize()
        return out[0].detach().clone(), [p.grad.clone() for p in P] + [xys_grad]

    try:
        ops.binning_reset()
        ops.set_segmented_backward(False)
        img0, g0 = run()
        ops.set_segmented_backward(True)
        img1, g1 = run()      # statistics of the frame above: planned
    finally:
        ops.set_segmented_backward(True)
    assert torch.equal(img0, img1)
    # The compositing-level gradients of the two schedules agree to ~1e-5 (the test above: a piece starts from
    # the forward's transmittance product, the one-pass walk from a product of up to a thousand reciprocals);
    # the projection backward multiplies that by the conditioning of the thinnest Gaussians (DESIGN.md §3,
    # "needles": up to ~100 x) — the same bound test_gpu_fused.py puts on two roundings of this chain
    names = ["means", "log_scales", "quats", "opacity_logits", "features_dc", "features_rest", "xys"]
    for n, a, b in zip(names, g0, g1):
        assert torch.isfinite(b).all(), n
        tol = 2e-5 if n in ("opacity_logits", "features_dc", "features_rest", "xys") else 2e-3
        assert rel_err(np_(b), np_(a)) < tol, n
    assert not all(torch.equal(a, b) for a, b in zip(g0, g1)), "the segmented backward did not run"


def test_rasterize_gaussians_operator_uses_the_pieces_from_its_second_frame_on():
    """The drop-in operator an unmodified model.cpp calls (rasterize_gaussians.hpp:23-37, ten arguments): from the
    second frame of a size on its backward runs in pieces; image the same bits, the four compositing-level
    gradients within summation order of the one-pass backward."""
    import torch

    from opensplat_amd import ops

    s = _deep_scene()    # random opacities up to 1: hot entries included
    v_img = to_dev(np.random.RandomState(5).uniform(-1, 1, (s.H, s.W, 3)).astype(np.float32))

    def run():
        t = lambda a, rg=False: to_dev(a).requires_grad_(rg)
        p = ops.project_gaussians(t(s.means), t(s.scales), 1.0, t(s.quats), t(s.viewmat), t(s.projmat),
                                  s.fx, s.fy, s.cx, s.cy, s.H, s.W)
        xys, conics = p[0].detach().requires_grad_(True), p[3].detach().requires_grad_(True)
        colors, opac = t(s.colors, True), t(s.opacities, True)
        img = ops.rasterize_gaussians(xys, p[1], p[2], conics, p[4], colors, opac, s.H, s.W, t(s.background),
                                      p[6])
        img.backward(v_img)
        torch.cuda.synchronize()
        return img.detach().clone(), [x.grad.clone() for x in (xys, conics, colors, opac)]

    try:
        ops.binning_reset()
        ops.set_segmented_backward(False)
        img0, g0 = run()
        ops.set_segmented_backward(True)
        img1, g1 = run()      # statistics of the frame above: planned
    finally:
        ops.set_segmented_backward(True)
    assert torch.equal(img0, img1)
    for n, a, b in zip(("xys", "conics", "colors", "opacity"), g0, g1):
        assert torch.isfinite(b).all(), n
        assert rel_err(np_(b), np_(a)) < 2e-5, n
    assert not all(torch.equal(a, b) for a, b in zip(g0, g1)), "the segmented backward did not run"
