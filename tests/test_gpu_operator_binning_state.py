"""-m gpu: the speculative-binning state of the C++ operators (torch_ops.cpp).

The id-list capacity RasterizeGaussians / SplatRender give the binning is the RUNNING MAXIMUM of
1.125 M + 1024 per (device, image width, image height): cameras visited in random order (what
OpenSplat's training loop does, opensplat.cpp:152) repeat a forward only while the maximum is still
being learnt, and training at 1/4 resolution does not disturb validation at full resolution.  The
reference blocks on cumsum().item() in every forward instead (rasterize_gaussians.cpp:62-63)."""
import numpy as np
import pytest

from opensplat_amd import scenes
from tests.util import np_, to_dev

pytestmark = pytest.mark.gpu

YAWS = [-21.0, -15.0, -9.0, -3.0, 3.0, 9.0, 15.0, 21.0]


def _render(ops, s, vm, pm, W, H, leaves=None):
    import torch

    t = lambda a: to_dev(a)
    fx = s.fx * W / s.W
    fy = s.fy * H / s.H
    p = ops.project_gaussians(t(s.means), t(s.scales), 1.0, t(s.quats), t(vm), t(pm), fx, fy, W / 2.0, H / 2.0,
                              H, W)
    rgb = torch.clamp_min(ops.spherical_harmonics(s.degrees_to_use, t(s.dirs), t(s.sh_coeffs)) + 0.5, 0.0)
    return ops.rasterize_gaussians(p[0], p[1], p[2], p[3], p[4], rgb, t(s.opacities), H, W, t(s.background), p[6])


def test_shuffled_cameras_repeat_a_forward_only_while_the_maximum_is_learnt(restated):
    import torch

    from opensplat_amd import ops

    # a scene whose intersection count depends strongly on the camera: a quarter of the Gaussians sits
    # in a window left of the centre, so that the yawed cameras see very different loads
    s = scenes.camera_scene(150_000, 960, 540, K=4, seed=31, sigma_px=(0.8, 5.0), hot=(0.25, 200))
    s.means[: s.N // 4, 0] -= 1.5
    cams = [scenes.yaw_camera(s.W, s.H, y) for y in YAWS]
    ops.binning_reset()
    dev = torch.cuda.current_device()
    rs = np.random.RandomState(0)
    order = rs.permutation(8)
    imgs = {}
    for c in order:                                   # pass 1: the maximum is being learnt
        imgs[c] = np_(_render(ops, s, *cams[c], s.W, s.H))
    calls1, repeats1 = ops.binning_counters()
    assert calls1 == 8 + repeats1 and repeats1 >= 1   # (the very first call always repeats: capacity 1024)
    cap = ops.binning_capacity(dev, s.W, s.H)
    for _ in range(3):                                # passes 2-4, other orders: nothing repeats
        for c in rs.permutation(8):
            img = np_(_render(ops, s, *cams[c], s.W, s.H))
            assert np.array_equal(img, imgs[c])       # (and a repeated forward would give the same image)
    calls2, repeats2 = ops.binning_counters()
    assert repeats2 == repeats1, (repeats1, repeats2)
    assert calls2 - calls1 == 24
    assert ops.binning_capacity(dev, s.W, s.H) == cap   # the maximum had been reached in pass 1
    # the intersection counts really differ by more than the 12.5 % margin between cameras: a hint that
    # followed the LAST frame (round 2) would have repeated forwards in every pass
    from opensplat_amd import cabi
    Ms = []
    for vm, pm in cams:
        camr = cabi.make_camera(vm, pm, s.fx, s.fy, s.cx, s.cy, s.W, s.H)
        p = cabi.project_forward(camr, to_dev(s.means), to_dev(s.scales), to_dev(s.quats))
        col = torch.zeros((s.N, 3), device="cuda")
        b = cabi.bin_and_sort(s.W, s.H, p["xys"], p["depths"], p["radii"], p["conics"], col,
                              to_dev(s.opacities.reshape(-1)), p["cov2d"])
        Ms.append(b.num_isects)
    assert max(Ms) > 1.25 * min(Ms), Ms
    assert cap == max(Ms) + max(Ms) // 8 + 1024


def test_image_sizes_keep_their_own_capacity(restated):
    """Training at a quarter of the resolution next to validation at full resolution
    (model.cpp:249-251): alternating sizes do not evict each other's capacity."""
    import torch

    from opensplat_amd import ops

    s = scenes.camera_scene(60_000, 640, 384, K=1, seed=32, sigma_px=(1.0, 6.0))
    vm, pm = scenes.yaw_camera(s.W, s.H, 0.0)
    ops.binning_reset()
    dev = torch.cuda.current_device()
    for W, H in [(640, 384), (160, 96), (640, 384), (160, 96)]:
        _render(ops, s, vm, pm, W, H)
    calls, repeats = ops.binning_counters()
    assert repeats == 2 and calls == 6                 # one cold start per size, none afterwards
    big, small = ops.binning_capacity(dev, 640, 384), ops.binning_capacity(dev, 160, 96)
    assert big > small > 0
    for W, H in [(160, 96), (640, 384)] * 3:
        _render(ops, s, vm, pm, W, H)
    assert ops.binning_counters() == (12, 2)
    assert ops.binning_capacity(dev, 640, 384) == big and ops.binning_capacity(dev, 160, 96) == small
    assert ops.binning_capacity(dev, 123, 45) == 0     # a size never rendered
