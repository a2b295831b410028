"""-m gpu: BASELINE config 3 (5 M Gaussians, 3840x2160, SH degree 3) on the WHOLE frame through
size-independent properties — the oracle needs minutes for a full C3 frame, so the value checks of this
size go through windows (tests/test_gpu_baseline_parity.py::test_c3_windows_match_oracle); what holds for
EVERY pixel and EVERY Gaussian of the timed path at that size is checked here:

  * the binning's contract: every tile's list sorted by depth, ids in range, every list entry's coverage
    mask non-contradictory with its Gaussian's rectangle, the lists' total equal to the reported M;
  * the compositing weights telescope: with every colour = 1 and background = 1 the image is 1 at every
    pixel, saturated or not (a walk that stops leaves the remaining transmittance to the background,
    gsplat_cpu.cpp:228-233) — to the rounding of a few hundred fp32 terms;
  * the backward is LINEAR in the cotangent: grad(a v1 + v2) = a grad(v1) + grad(v2) for all six tensors;
  * the background enters as T_final * bg: two backgrounds differ by exactly that image, and d/d(bg) of
    the loss <v, img> shows up nowhere in the parameter gradients except through T (checked by linearity).
"""
import numpy as np
import pytest

from opensplat_amd import scenes
from tests.test_gpu_baseline_parity import run_timed_path, timed_path_grads
from tests.util import np_, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c3():
    s = scenes.config_c3()
    return s, run_timed_path(s)


def test_c3_binning_contract_on_the_whole_frame(c3):
    import torch

    s, pipe = c3
    b = pipe.ws
    tiles = ((s.W + 15) // 16) * ((s.H + 15) // 16)
    bins = b.bufs["tile_bins"].reshape(-1)[: 2 * tiles].view(tiles, 2).long()
    M = int(pipe.num_isects)
    ids = b.bufs["ids_sorted"].reshape(-1)[:M].long()
    masks = b.bufs["block_masks"].reshape(-1)[:M].long() & 0xFFFF
    lens = bins[:, 1] - bins[:, 0]
    assert int(lens.sum()) == M and int(lens.min()) >= 0
    assert bool((bins[1:, 0] == bins[:-1, 1]).all()) and int(bins[0, 0]) == 0 and int(bins[-1, 1]) == M
    assert int(ids.min()) >= 0 and int(ids.max()) < s.N
    # depth-sorted inside every tile: the depth of entry i+1 is >= the one of entry i unless a tile starts there
    depth = pipe.proj["depths"][ids]
    starts = torch.zeros(M, dtype=torch.bool, device=ids.device)
    starts[bins[lens > 0, 0]] = True
    bad = (depth[1:] < depth[:-1]) & ~starts[1:]
    assert int(bad.sum()) == 0
    # every entry: a visible Gaussian whose pixel rectangle meets the tile; its mask only names blocks the
    # rectangle reaches (the packed record carries the rectangle: words 7 and 11)
    pk = pipe.gfwd["packed"]
    rx = pk[:, 7].view(torch.int32)[ids].long() & 0xFFFFFFFF
    ry = pk[:, 11].view(torch.int32)[ids].long() & 0xFFFFFFFF
    x0, x1, y0, y1 = rx & 0xFFFF, rx >> 16, ry & 0xFFFF, ry >> 16
    tile = torch.repeat_interleave(torch.arange(tiles, device=ids.device), lens)
    tiles_x = (s.W + 15) // 16
    tx0, ty0 = (tile % tiles_x) * 16, (tile // tiles_x) * 16
    assert bool(((x1 > tx0) & (x0 < tx0 + 16) & (y1 > ty0) & (y0 < ty0 + 16)).all())
    assert int((pipe.proj["radii"][ids] <= 0).sum()) == 0
    for r in range(4):
        for c in range(4):
            bit = (masks >> (4 * r + c)) & 1
            reach = (x1 > tx0 + 4 * c) & (x0 < tx0 + 4 * c + 4) & (y1 > ty0 + 4 * r) & (y0 < ty0 + 4 * r + 4)
            assert int((bit.bool() & ~reach).sum()) == 0, (r, c)
    assert float((masks != 0).float().mean()) > 0.9      # (a few per cent of the entries are empty: DESIGN §9)


def test_c3_compositing_weights_telescope_to_one(c3):
    """colour = 1 everywhere (dc = 0.5 / C0, rest = 0), background = 1: sum_i alpha_i T_i + T_final = 1."""
    import torch

    s, pipe = c3
    dc, rest, bg = pipe.features_dc.clone(), pipe.features_rest.clone(), pipe.background.clone()
    try:
        pipe.features_dc.fill_(0.5 / 0.28209479177387814)
        pipe.features_rest.zero_()
        pipe.background.fill_(1.0)
        pipe.step()
        torch.cuda.synchronize()
        img = pipe.fwd["img"]
        assert float((pipe.gfwd["packed"][:, 8:11][pipe.proj["radii"] > 0] - 1.0).abs().max()) < 2e-7
        # (a walk that stops hands the transmittance IN FRONT of the Gaussian it drops to the background,
        # gsplat_cpu.cpp:228-233: the sum telescopes for saturated pixels too — 92 % of this frame)
        err = (img - 1.0).abs()
        print("telescoping: max |img - 1| = %.3g, mean %.3g, saturated pixels %.3f" % (
            float(err.max()), float(err.mean()), float((pipe.fwd["final_Ts"] < 1.0e-3).float().mean())))
        assert float(err.max()) < 5e-6, float(err.max())          # (measured 8e-7)
    finally:
        pipe.features_dc.copy_(dc); pipe.features_rest.copy_(rest); pipe.background.copy_(bg)
        pipe.step()
        torch.cuda.synchronize()


def test_c3_backward_is_linear_in_the_cotangent_and_background_enters_through_t(c3):
    import torch

    s, pipe = c3
    v0 = pipe.v_out.clone()
    rs = np.random.RandomState(11)
    v1 = to_dev(rs.uniform(-1, 1, (s.H, s.W, 3)).astype(np.float32))
    a = 0.375                                                   # (exact in binary: a*v1 is exact scaling)

    def grads_for(v):
        pipe.v_out.copy_(v)
        pipe.step()
        torch.cuda.synchronize()
        return timed_path_grads(pipe)
    try:
        g0, g1, g01 = grads_for(v0), grads_for(v1), grads_for(a * v1 + v0)
        for k in g0:
            want = a * g1[k].astype(np.float64) + g0[k]
            err = np.abs(g01[k] - want).max() / max(np.abs(want).max(), 1e-30)
            assert err < 1e-5, (k, err)
        # the image of another background = this image + T_final * (bg' - bg), to rounding
        img0, T = np_(pipe.fwd["img"]), np_(pipe.fwd["final_Ts"])
        bg0 = np_(pipe.background).copy()
        pipe.background.copy_(to_dev(np.array([0.9, 0.1, 0.6], np.float32)))
        pipe.step()
        torch.cuda.synchronize()
        img1 = np_(pipe.fwd["img"])
        assert np.array_equal(np_(pipe.fwd["final_Ts"]), T)
        want = img0.astype(np.float64) + T[..., None] * (np.array([0.9, 0.1, 0.6]) - bg0)
        assert np.abs(img1 - want).max() < 3e-7
        pipe.background.copy_(to_dev(bg0))
    finally:
        pipe.v_out.copy_(v0)
        pipe.step()
        torch.cuda.synchronize()
