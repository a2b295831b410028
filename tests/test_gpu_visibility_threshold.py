"""-m gpu: the alpha >= 1/255 decision at the threshold itself.

The compositing backward decides "did the forward composite this (pixel, Gaussian) pair" on
vis = exp(-sigma) against a per-entry threshold (1/255) / opacity with a +-4e-6 band, and redoes the
forward's exact arithmetic (its sigma, the glibc-exact exponential, alpha = opacity * vis >= 1/255,
gsplat_cpu.cpp:218-224) inside the band (gs_raster.hip, SRecB).  A random scene puts ~1e-6 of its pairs
into the band; this scene puts EVERY pair there: Gaussians a few ulps above and below the threshold, at
sigma = 0 (on a pixel) and at sigma > 0, each alone on its pixels.  The backward must give a gradient
to exactly the pairs the oracle's forward composited."""
import numpy as np
import pytest

from opensplat_amd import cabi
from tests.util import np_, to_dev

pytestmark = pytest.mark.gpu


def _scene(W=192, H=128, seed=0):
    rs = np.random.RandomState(seed)
    thr = np.float32(1.0) / np.float32(255.0)
    gx, gy = np.meshgrid(np.arange(6, W - 6, 6), np.arange(6, H - 6, 6))
    n = gx.size
    # half of the Gaussians exactly on a pixel centre (sigma = 0 there), half offset by a random fraction
    off = np.where(rs.rand(n, 1) < 0.5, 0.0, rs.uniform(-0.45, 0.45, (n, 2)))
    # (the CPU rasterizer's pixel (i, j) sits AT (j, i): xCam = gX - j, gsplat_cpu.cpp:211-212)
    xys = (np.stack([gx.ravel(), gy.ravel()], -1) + off).astype(np.float32)
    # isotropic conics: only the nearest pixel can come near the threshold (the next one has a sigma at
    # least 0.2 larger: alpha <= 0.82 / 255)
    a = rs.uniform(4.0, 10.0, n).astype(np.float32)
    conics = np.stack([a, np.zeros(n, np.float32), a], -1)
    cov2d = np.stack([1.0 / a, np.zeros(n, np.float32), 1.0 / a], -1).astype(np.float32)
    # opacity such that opacity * exp(-sigma_nearest) sits within a few ulps of 1/255
    d = xys - np.round(xys)
    sig = (0.5 * (a * d[:, 0] * d[:, 0] + a * d[:, 1] * d[:, 1])).astype(np.float64)
    opac = (np.float64(thr) * np.exp(sig)).astype(np.float32)
    ulps = rs.randint(-6, 7, n)
    for _ in range(6):
        opac = np.where(ulps > 0, np.nextafter(opac, np.float32(2)), np.where(ulps < 0, np.nextafter(opac, np.float32(0)), opac))
        ulps = ulps - np.sign(ulps)
    opac = np.minimum(opac, np.float32(0.99)).astype(np.float32)
    colors = rs.uniform(0.2, 1.0, (n, 3)).astype(np.float32)
    depths = rs.uniform(1.0, 5.0, n).astype(np.float32)
    radii = np.full(n, 2, np.int32)
    return dict(W=W, H=H, xys=xys, conics=conics, cov2d=cov2d, opac=opac, colors=colors, depths=depths, radii=radii,
                bg=np.zeros(3, np.float32), v_out=rs.uniform(0.5, 1.0, (H, W, 3)).astype(np.float32))


@pytest.mark.parametrize("seed", [0, 1])
def test_backward_takes_the_forwards_decision_at_the_threshold(seed, restated):
    import torch

    s = _scene(seed=seed)
    W, H, n = s["W"], s["H"], len(s["xys"])
    b = cabi.bin_and_sort(W, H, to_dev(s["xys"]), to_dev(s["depths"]), to_dev(s["radii"]), to_dev(s["conics"]),
                          to_dev(s["colors"]), to_dev(s["opac"]), to_dev(s["cov2d"]))
    f = cabi.rasterize_forward(W, H, b, s["bg"], 0)
    g = cabi.rasterize_backward(W, H, n, b, s["bg"], f["final_Ts"], f["final_idx"], to_dev(s["v_out"]), 0)
    torch.cuda.synchronize()
    c2 = np.zeros((n, 2, 2), np.float32)
    c2[:, 0, 0], c2[:, 1, 1] = s["cov2d"][:, 0], s["cov2d"][:, 2]
    of = restated.rasterize_forward(W, H, s["xys"], s["conics"], s["colors"], s["opac"], s["bg"], c2, s["depths"])
    og = restated.rasterize_backward(W, H, s["xys"], s["conics"], s["colors"], s["opac"], s["bg"], c2, s["depths"],
                                     of["final_Ts"], of["state"], s["v_out"])
    assert np.array_equal(np_(f["img"]), of["img"]) and np.array_equal(np_(f["final_Ts"]), of["final_Ts"])
    # the scene does what it is for: a good part of the Gaussians is composited, a good part is not
    drawn = og["v_opacity"] != 0
    assert 0.2 * n < drawn.sum() < 0.8 * n, drawn.sum()
    # the same pairs get a gradient (a disagreement would show as a gradient ~1 where the oracle has 0)
    assert np.array_equal(np_(g["v_opacity"]) != 0, drawn)
    for k in ("v_xy", "v_conic", "v_colors", "v_opacity"):
        err = np.abs(np_(g[k]).astype(np.float64) - og[k]).max() / max(np.abs(og[k]).max(), 1e-30)
        assert err < 2e-6, (k, err)
